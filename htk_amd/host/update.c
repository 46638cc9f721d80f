/* update.c -- host-side (C) model update after a Baum-Welch pass: MLUpdateModels of HERest.
 *
 * The update is a few milliseconds of scalar work over the whole HMM set and stays on the host, as in the
 * reference.  Input is the fp64 accumulator vector the device produced (htkamd_accs layout); each entry is
 * rounded to float ONCE and from there the arithmetic follows the reference's float expressions:
 *   UpdateTrans   HERest.c:795-816    a_ij = tran/occ -> log
 *   UpdateWeights HERest.c:897-971    c_m/occ, > MINMIX else 0, FloorMixes HERest.c:819-840
 *   UpdateVars    HERest.c:1045-1122  va/occ - (mu/occ)^2, floored; BEFORE the means
 *   UpdateMeans   HERest.c:974-1012   mean += mu/occ
 *   FixGConsts    HModel.c:5688-5714
 *   MLUpdateModels HERest.c:1262-1321 models with fewer than minEgs examples are left alone; a shared
 *                 structure (tied state, Gaussian, transP) is updated by the first model that qualifies.
 * Tied mean / variance vectors (~u / ~v; htkamd_model_set_sharing): the reference hangs ONE MuAcc / VaAcc on a shared vector, so the
 * statistics of its users are pooled (here: summed into the group's first Gaussian before the update), the vector is updated by the
 * first mixture that reaches it, a variance with several users gets no mean-shift correction (`shared`, HERest.c:1080), and a private
 * variance whose tied mean was already moved by an earlier model gets none either (its MuAcc hook is gone by then).
 * MAP (HTKAMD_UPMAP, HERest -u p...): MAPUpdateModels HMap.c:413-460 from the same accumulators --
 *   UpdateWeights HMap.c:205-276   (max(0, w*vSize*tau - 1) + c_m) / (sum of those + occ), no MINMIX zeroing, forced renormalisation
 *   UpdateVars    HMap.c:314-377   (tau*var + va - muDiff) / (tau + occ), muDiff = 2 dmu mu - dmu^2 occ, dmu = mu / (tau + occ)
 *   UpdateMeans   HMap.c:279-311   (mean*tau + (mu + mean*occ)) / (tau + occ)
 * singleProcess mirrors parMode == -1 (HERest.c:1336-1339): the set went through ConvDiagC/ConvLogWt before
 * the pass, so parameters that are not re-estimated come back through ForceDiagC/ConvExpWt float round trips.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

int htkamd_update_models(struct htkamd_model *m, const htkamd_accs_layout *lay, const double *acc,
                         const htkamd_update_config *cfg, htkamd_update_stats *st)
{
   const int D = m->D;
   int h, i, j, k, c, maxM = m->maxM;
   unsigned char *doneT, *doneS, *doneMu, *doneVa;
   float *mean = m->h_mean, *var = m->h_var, *gconst = m->h_gconst, *wgt = m->h_compWeight, *transP = m->h_transP;
   if (!m || !lay || !acc || !cfg || !st) { htkamd_set_error("update_models: NULL argument"); return HTKAMD_EINVAL; }
   memset(st, 0, sizeof(*st));
   const int map = (cfg->uFlags & HTKAMD_UPMAP) != 0;
   const float mapTau = cfg->mapTau;
   if (map && m->NSt > 1) { htkamd_set_error("update_models: MAP re-estimation of a multi-stream set is not supported"); return HTKAMD_EMODEL; }
   if (map && (cfg->uFlags & HTKAMD_UPTRANS)) { htkamd_set_error("update_models: no MAP update of transition probabilities (HMap.c:434, HError 999)"); return HTKAMD_EINVAL; }
   doneT = (unsigned char *)calloc((size_t)m->nT, 1);
   doneS = (unsigned char *)calloc((size_t)m->S, 1);
   doneMu = (unsigned char *)calloc((size_t)m->G, 1);
   doneVa = (unsigned char *)calloc((size_t)m->G, 1);
   const int *mL = m->h_meanLeader, *vL = m->h_varLeader;
   double *pool = NULL;
   if (mL) {                                            /* one accumulator per shared vector */
      int g;
      pool = (double *)malloc(sizeof(double) * lay->total);
      memcpy(pool, acc, sizeof(double) * lay->total);
      for (g = 0; g < m->G; g++) {
         if (mL[g] != g) {
            for (k = 0; k < D; k++) pool[lay->mu + (size_t)mL[g] * D + k] += acc[lay->mu + (size_t)g * D + k];
            pool[lay->muOcc + mL[g]] += acc[lay->muOcc + g];
         }
         if (vL[g] != g) {
            for (k = 0; k < D; k++) pool[lay->va + (size_t)vL[g] * D + k] += acc[lay->va + (size_t)g * D + k];
            pool[lay->vaOcc + vL[g]] += acc[lay->vaOcc + g];
         }
      }
      acc = pool;
   }
/* several streams: a Gaussian lives on the dimensions of its stream, the rest of its row is never touched */
#define OUTSIDE(g, k) (m->NSt > 1 && m->h_dimStream[k] != m->h_gaussStream[g])
#define ML(g) (mL ? mL[g] : (g))
#define VL(g) (vL ? vL[g] : (g))
#define ACCF(off, idx) ((float)acc[(off) + (size_t)(idx)])
   if (m->tiedMix) {
      /* hsKind TIEDHS (MLUpdateModels HERest.c:1272-1279): the pool is re-estimated ONCE per set, before the models are visited and
         whatever their example counts -- UpdateTMVars (:1124-1195), UpdateTMMeans (:1014-1042), FixAllGConsts; a pool Gaussian has no
         weight to test, its variance carries the mean-shift term whenever its mean has statistics */
      int ks, g;
      if (map) { htkamd_set_error("update_models: MAP re-estimation of a tied-mixture set is not supported"); free(doneT); free(doneS); free(doneMu); free(doneVa); free(pool); return HTKAMD_EMODEL; }
      if (cfg->uFlags & HTKAMD_UPVARS)
         for (ks = 0; ks < m->NSt; ks++)
            for (c = m->h_stateCompOff[ks]; c < m->h_stateCompOff[ks + 1]; c++) {
               g = m->h_compGauss[c];
               const float occim = ACCF(lay->vaOcc, g), muOcc = ACCF(lay->muOcc, g);
               int mixFloored = 0;
               if (!(occim > 0.0)) continue;
               const int shared = ((cfg->uFlags & HTKAMD_UPMEANS) == 0 || muOcc <= 0.0);
               for (k = 0; k < D; k++) {
                  if (OUTSIDE(g, k)) continue;
                  const float muDiffk = shared ? 0.0 : ACCF(lay->mu, (size_t)g * D + k) / muOcc;
                  float x = ACCF(lay->va, (size_t)g * D + k) / occim - muDiffk * muDiffk;
                  const float fl = cfg->varFloor ? cfg->varFloor[k] : cfg->minVar;
                  if (x < fl) { x = fl; st->nFloorVar++; mixFloored = 1; }
                  var[(size_t)g * D + k] = x;
               }
               if (mixFloored) st->nFloorVarMix++;
            }
      if (cfg->uFlags & HTKAMD_UPMEANS)
         for (ks = 0; ks < m->NSt; ks++)
            for (c = m->h_stateCompOff[ks]; c < m->h_stateCompOff[ks + 1]; c++) {
               g = m->h_compGauss[c];
               const float occim = ACCF(lay->muOcc, g);
               if (occim > 0.0) for (k = 0; k < D; k++) { if (OUTSIDE(g, k)) continue; mean[(size_t)g * D + k] += ACCF(lay->mu, (size_t)g * D + k) / occim; }
            }
      if (cfg->uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS))
         for (g = 0; g < m->G; g++) htkamd_host_fix_diag_gconst_ms(D, var + (size_t)g * D, m->h_dimStream, m->h_gaussStream ? m->h_gaussStream[g] : 0, gconst + g);
   }
   if (cfg->singleProcess && !m->tiedMix) {             /* ConvDiagC / ConvLogWt leave a TIEDHS set alone (HUtil.c:419,478) */
      /* the conversions walk the set with an HMM scan: a state macro that no model uses keeps its values */
      unsigned char *usedS = (unsigned char *)calloc((size_t)m->S, 1), *usedG = (unsigned char *)calloc((size_t)m->G, 1);
      size_t z;
      int g, s0;
      for (h = 0; h < m->H; h++)
         for (j = m->h_hmmStateOff[h]; j < m->h_hmmStateOff[h + 1]; j++) usedS[m->h_hmmState[j]] = 1;
      for (s0 = 0; s0 < m->S; s0++)
         if (usedS[s0]) for (c = m->h_stateCompOff[s0]; c < m->h_stateCompOff[s0 + 1]; c++) usedG[m->h_compGauss[c]] = 1;
      for (g = 0; g < m->G; g++) {
         if (!usedG[g]) continue;
         for (z = (size_t)g * D; z < (size_t)(g + 1) * D; z++) {
            float v = var[z], iv;
            if (OUTSIDE(g, (int)(z - (size_t)g * D))) continue;
            if (v > 1E+30) v = 1E+30;
            if (v < 1E-30) v = 1E-30;
            iv = 1 / v;
            if (iv > 1E+30) iv = 1E+30;
            if (iv < 1E-30) iv = 1E-30;
            var[z] = 1 / iv;
         }
      }
      for (s0 = 0; s0 < m->S; s0++)
         if (usedS[s0]) for (c = m->h_stateCompOff[s0]; c < m->h_stateCompOff[s0 + 1]; c++)
            if (!(m->h_rawLogWt && m->h_rawLogWt[c])) wgt[c] = exp(htkamd_host_mix_log_weight(wgt[c]));      /* (a weight ConvLogWt skipped, ConvExpWt skips too) */
      free(usedS); free(usedG);
   }
   /* UpdateModels walks the set with an HMM scan (HERest.c:1262-1321: NewHMMScan / GoNextHMM, the hash order of the physical models'
      names).  For private parameters the order is immaterial; for a mean shared by Gaussians of several models it decides whose variance
      carries the mean-shift term (the first mixture to reach the shared mean: `shared` below).  h_scanOrder (htkamd_model_set_scan_order)
      is that order; without it the models are taken as defined. */
   for (int hh = 0; hh < m->H; hh++) {
      h = m->h_scanOrder ? m->h_scanOrder[hh] : hh;
      const int n = (int)llround(acc[lay->nEgs + h]), ti = m->h_hmmTrans[h], N = m->h_transN[ti];
      const int *hs = m->h_hmmState + m->h_hmmStateOff[h];
      const int nHs = m->h_hmmStateOff[h + 1] - m->h_hmmStateOff[h];          /* N - 2 states, or their (state, stream) elements */
      if (n < cfg->minEgs) st->nSkippedHmm++;
      if (!(n >= cfg->minEgs && n > 0)) continue;
      if ((cfg->uFlags & HTKAMD_UPTRANS) && !doneT[ti]) {
         float *tp = transP + m->h_transOff[ti];
         for (i = 1; i < N; i++) {
            const float occi = ACCF(lay->trOcc, m->h_trOccOff[ti] + i - 1);
            if (occi > 0.0 && cfg->rowNormalise) {       /* HRest's RestTransP (HRest.c:1015-1039): the row is renormalised by its sum */
               float sum = 0.0, row[N + 1];
               for (j = 2; j <= N; j++) { row[j - 1] = ACCF(lay->tr, m->h_transOff[ti] + (i - 1) * N + (j - 1)) / occi; sum += row[j - 1]; }
               for (j = 2; j <= N; j++) {
                  const float x = row[j - 1] / sum;
                  tp[(i - 1) * N + (j - 1)] = (x < MINLARG) ? LZERO : log(x);
               }
            } else if (occi > 0.0)
               for (j = 2; j <= N; j++) {
                  const float x = ACCF(lay->tr, m->h_transOff[ti] + (i - 1) * N + (j - 1)) / occi;
                  tp[(i - 1) * N + (j - 1)] = (x > MINLARG) ? log(x) : LZERO;
               }
            else st->nNoTransOut++;
         }
         doneT[ti] = 1;
      }
      if (maxM > 1 && (cfg->uFlags & HTKAMD_UPMIXES))
         for (j = 0; j < nHs; j++) {
            const int s = hs[j], c0 = m->h_stateCompOff[s], M = m->h_stateCompOff[s + 1] - c0;
            const float occi = ACCF(lay->wtOcc, s);
            if (doneS[s]) continue;
            if (occi > 0 && map) {                        /* HMap.c:227-268 */
               float denom = 0, x;
               for (k = 0; k < M; k++) {
                  float tmp = wgt[c0 + k] * D * mapTau - 1;
                  if (tmp < 0) tmp = 0;
                  denom += tmp;
               }
               for (k = 0; k < M; k++) {
                  float tmp = wgt[c0 + k] * D * mapTau - 1;
                  if (tmp < 0) tmp = 0;
                  x = (tmp + ACCF(lay->wt, c0 + k)) / (denom + occi);
                  if (x > 1.001) st->nWeightAboveOne++;
                  if (x > 1.0) x = 1.0;
                  wgt[c0 + k] = x;
               }
               if (cfg->mixWeightFloor > 0.0) {
                  float sum = 0.0, fsum = 0.0, scale;
                  const float floor = cfg->mixWeightFloor;
                  for (k = 0; k < M; k++) {
                     if (wgt[c0 + k] > floor) sum += wgt[c0 + k];
                     else { fsum += floor; wgt[c0 + k] = floor; }
                  }
                  if (fsum != 0.0 && sum != 0.0) {
                     scale = (1.0 - fsum) / sum;
                     for (k = 0; k < M; k++)
                        if (wgt[c0 + k] > floor) wgt[c0 + k] *= scale;
                  }
               }
               x = 0;                                    /* "Force a normalisation becomes of weird zeroing" */
               for (k = 0; k < M; k++) x += wgt[c0 + k];
               for (k = 0; k < M; k++) wgt[c0 + k] /= x;
            } else if (occi > 0) {
               for (k = 0; k < M; k++) {
                  float x = ACCF(lay->wt, c0 + k) / occi;
                  if (x > 1.001) st->nWeightAboveOne++;          /* fatal HError 2393 in the reference (HERest.c:926) */
                  if (x > 1.0) x = 1.0;
                  wgt[c0 + k] = (x > MINMIX) ? x : 0.0;
               }
               if (cfg->mixWeightFloor > 0.0) {
                  float sum = 0.0, fsum = 0.0, scale;
                  const float floor = cfg->mixWeightFloor;
                  for (k = 0; k < M; k++) {
                     if (wgt[c0 + k] > floor) sum += wgt[c0 + k];
                     else { fsum += floor; wgt[c0 + k] = floor; }
                  }
                  if (fsum != 0.0 && sum != 0.0) {
                     scale = (1.0 - fsum) / sum;
                     for (k = 0; k < M; k++)
                        if (wgt[c0 + k] > floor) wgt[c0 + k] *= scale;
                  }
               }
            } else if (!map) st->nNoMixUse++;
            doneS[s] = 1;
         }
      if (m->tiedMix) continue;                          /* the pool was done above */
      if (cfg->uFlags & HTKAMD_UPVARS)
         for (j = 0; j < nHs; j++) {
            const int s = hs[j];
            for (c = m->h_stateCompOff[s]; c < m->h_stateCompOff[s + 1]; c++)
               if (wgt[c] > MINMIX) {
                  const int g0 = m->h_compGauss[c], g = VL(g0), gm = ML(g0);
                  if (!doneVa[g]) {
                     const float occim = ACCF(lay->vaOcc, g);
                     int mixFloored = 0;
                     if (occim > 0.0) {
                        const float muOcc = ACCF(lay->muOcc, gm);
                        const int shared = ((cfg->uFlags & HTKAMD_UPMEANS) == 0 || doneMu[gm] || muOcc <= 0.0 || (vL && m->h_varGroupSize[g] > 1));
                        for (k = 0; k < D; k++) {
                           if (OUTSIDE(g, k)) continue;
                           float muDiffk = shared ? 0.0 : ACCF(lay->mu, (size_t)gm * D + k) / muOcc;
                           float x = ACCF(lay->va, (size_t)g * D + k) / occim - muDiffk * muDiffk;
                           if (map) {                    /* HMap.c:350-356 */
                              if (shared) muDiffk = 0.0;
                              else {
                                 const float muk = ACCF(lay->mu, (size_t)gm * D + k), dmu = muk / (mapTau + occim);
                                 muDiffk = 2 * dmu * muk - dmu * dmu * occim;
                              }
                              x = (mapTau * var[(size_t)g * D + k] + ACCF(lay->va, (size_t)g * D + k) - muDiffk) / (mapTau + occim);
                           }
                           const float fl = cfg->varFloor ? cfg->varFloor[k] : cfg->minVar;
                           if (x < fl) { x = fl; st->nFloorVar++; mixFloored = 1; }
                           var[(size_t)g * D + k] = x;
                        }
                     } else if (!map) st->nNoVarUse++;
                     if (mixFloored) st->nFloorVarMix++;
                     doneVa[g] = 1;
                  }
               }
         }
      if (cfg->uFlags & HTKAMD_UPMEANS)
         for (j = 0; j < nHs; j++) {
            const int s = hs[j];
            for (c = m->h_stateCompOff[s]; c < m->h_stateCompOff[s + 1]; c++)
               if (wgt[c] > MINMIX) {
                  const int g = ML(m->h_compGauss[c]);
                  if (!doneMu[g]) {
                     const float occim = ACCF(lay->muOcc, g);
                     if (occim > 0.0 && map) {             /* HMap.c:301-304 */
                        if (occim > cfg->mapMinObs) st->nMapObserved++;
                        for (k = 0; k < D; k++) {
                           float *mk = mean + (size_t)g * D + k;
                           *mk = (*mk * mapTau + (ACCF(lay->mu, (size_t)g * D + k) + *mk * occim)) / (mapTau + occim);
                        }
                     } else if (occim > 0.0)
                        for (k = 0; k < D; k++) { if (OUTSIDE(g, k)) continue; mean[(size_t)g * D + k] += ACCF(lay->mu, (size_t)g * D + k) / occim; }
                     doneMu[g] = 1;
                  }
               }
         }
      if (cfg->uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS))
         for (j = 0; j < nHs; j++) {
            const int s = hs[j];
            for (c = m->h_stateCompOff[s]; c < m->h_stateCompOff[s + 1]; c++)
               if (wgt[c] > MINMIX) {
                  const int g = m->h_compGauss[c];
                  htkamd_host_fix_diag_gconst_ms(D, var + (size_t)VL(g) * D, m->h_dimStream, m->h_gaussStream ? m->h_gaussStream[g] : 0, gconst + g);
               }
         }
   }
#undef ACCF
   if (mL) {                                            /* every user of a shared vector sees its new value */
      int g;
      for (g = 0; g < m->G; g++) {
         if (mL[g] != g) memcpy(mean + (size_t)g * D, mean + (size_t)mL[g] * D, sizeof(float) * (size_t)D);
         if (vL[g] != g) memcpy(var + (size_t)g * D, var + (size_t)vL[g] * D, sizeof(float) * (size_t)D);
      }
   }
#undef ML
#undef OUTSIDE
#undef VL
   free(pool);
   free(doneT); free(doneS); free(doneMu); free(doneVa);
   if (st->nWeightAboveOne > 0) {
      htkamd_set_error("update_models: %d mixture weights above 1.001 (HERest: HError 2393): corrupt accumulators?", st->nWeightAboveOne);
      return HTKAMD_EMODEL;
   }
   return HTKAMD_OK;
}
