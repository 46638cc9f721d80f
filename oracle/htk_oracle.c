/* htk_oracle.c -- CPU restatement of the reference HTK hot path (TEST INFRASTRUCTURE).
 * See htk_oracle.h.  Compile WITHOUT FMA contraction (-ffp-contract=off) and without -ffast-math:
 * the reference is SSE2 scalar float, FLT_EVAL_METHOD == 0.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "htk_oracle.h"

/* ------------------------------------------------------------------ log arithmetic */

/* HMath.c:1680  minLogExp = -log(-LZERO) */
static double orc_min_log_exp(void)
{
   static double v = 0.0;
   if (v == 0.0) v = -log(-ORC_LZERO);
   return v;
}

/* HMath.c:1576-1590 LAdd */
double orc_ladd(double x, double y)
{
   double temp, diff, z;
   if (x < y) { temp = x; x = y; y = temp; }
   diff = y - x;
   if (diff < orc_min_log_exp())
      return (x < ORC_LSMALL) ? ORC_LZERO : x;
   z = exp(diff);
   return x + log(1.0 + z);
}

/* ------------------------------------------------------------------ model preparation */

/* HModel.c:5641-5654 FixDiagGConst: float sum, log evaluated in double, each term rounded to float */
void orc_fix_diag_gconst(int D, const float *var, float *gconst_out)
{
   float sum, z;
   int i;
   sum = D * log(ORC_TPI);
   for (i = 0; i < D; i++) {
      z = (var[i] <= ORC_MINLARG) ? ORC_LZERO : log(var[i]);
      sum += z;
   }
   *gconst_out = sum;
}

void orc_fix_diag_gconst_ms(int D, const float *var, const int *dimStream, int stream, float *gconst_out)
{
   float sum, z;
   int i, n = 0;
   for (i = 0; i < D; i++) if (!dimStream || dimStream[i] == stream) n++;
   sum = n * log(ORC_TPI);                 /* vSize = the stream's width */
   for (i = 0; i < D; i++) {
      if (dimStream && dimStream[i] != stream) continue;
      z = (var[i] <= ORC_MINLARG) ? ORC_LZERO : log(var[i]);
      sum += z;
   }
   *gconst_out = sum;
}

/* HUtil.c:413-437 ConvDiagC(convData=TRUE): clamp to [MINVAR,MAXVAR] then 1/v in float */
void orc_conv_diagc(int n, const float *var, float *ivar_out)
{
   int k;
   for (k = 0; k < n; k++) {
      float v = var[k];
      if (v > 1E+30) v = 1E+30;
      if (v < 1E-30) v = 1E-30;
      ivar_out[k] = 1 / v;
   }
}

/* HModel.c:5288-5295 MixLogWeight with hset->logWt == FALSE */
float orc_mix_log_weight(float w)
{
   if (w < ORC_MINMIX) return ORC_LZERO;
   return log(w);
}

/* HFB.c:91-102 FindStateOrder + HFB.c:106-155 SetMinDurs for one transition matrix.
   tp is row-major N*N with 1-based state numbering mapped to tp[(i-1)*N + (j-1)]. */
#define TP(i,j) tp[((i)-1)*N + ((j)-1)]
static void find_state_order(int N, const float *tp, int *so, int s, int *d)
{
   int p;
   so[s] = 0;
   for (p = 1; p < N; p++)
      if (TP(p, s) > ORC_LSMALL && p != s)
         if (so[p] < 0) find_state_order(N, tp, so, p, d);
   so[s] = ++(*d);
}

int orc_min_dur(int N, const float *tp)
{
   int *md = (int *)malloc(sizeof(int) * (N + 1));
   int *so = (int *)malloc(sizeof(int) * (N + 1));
   int i, j, k, d, nDS = 0, res;
   for (i = 1; i <= N; i++) so[i] = md[i] = -1;
   find_state_order(N, tp, md, N, &nDS);
   for (i = 1; i <= nDS; i++) so[md[i]] = i;
   for (i = 1; i <= N; i++) md[i] = N;
   for (k = 1, md[1] = 0; k <= nDS; k++) {
      i = so[k];
      if (i < 1 || i > N) continue;
      for (j = 1; j < N; j++)
         if (TP(j, i) > ORC_LSMALL) {
            d = md[j] + ((i == N) ? 0 : 1);
            if (d < md[i]) md[i] = d;
         }
   }
   if (md[N] < 0 || md[N] >= N)
      res = (TP(1, N) > ORC_LSMALL) ? 0 : 1;   /* HFB.c:144-149 */
   else
      res = md[N];
   free(md); free(so);
   return res;
}
#undef TP

/* ------------------------------------------------------------------ GMM scoring */

/* HModel.c:5420-5431 IDOutP: float sequential sum; -0.5*sum is exact */
float orc_idoutp(const float *x, int D, const float *mean, const float *ivar, float gconst)
{
   int i;
   float sum, xmm;
   sum = gconst;
   for (i = 0; i < D; i++) {
      xmm = x[i] - mean[i];
      sum += xmm * xmm * ivar[i];
   }
   return -0.5 * sum;
}

/* HFB.c:898-988 ShStrP (no shared-mix PreComp cache, no PDE, xform==NULL so det==0) and
   HRec.c:438-510 cSOutP: LogFloat x,mixp,wt -> float rounding after every LAdd. */
float orc_state_outp(const orc_model *m, int s, const float *x, float *mixp_out)
{
   int c0 = m->stateCompOff[s], c1 = m->stateCompOff[s + 1], M = c1 - c0, k, g;
   float xx, mixp, wt, det = 0.0f;
   if (M == 1) {                         /* HFB.c:917-928 */
      g = m->compGauss[c0];
      xx = orc_idoutp(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g]);
      xx += det;
      if (mixp_out) mixp_out[0] = xx;
      return xx;
   }
   xx = ORC_LZERO;                       /* HFB.c:949-960 */
   for (k = 0; k < M; k++) {
      if (mixp_out) mixp_out[k] = ORC_LZERO;      /* NewOtprobVec HFB.c:883-894 */
      wt = m->compLogWt[c0 + k];
      if (wt > ORC_LMINMIX) {
         g = m->compGauss[c0 + k];
         mixp = orc_idoutp(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g]);
         mixp += det;
         xx = orc_ladd(xx, wt + mixp);   /* float add, double LAdd, float store */
         if (mixp_out) mixp_out[k] = mixp;
      }
   }
   return xx;
}

/* IDOutP over the dimensions of one stream of the undivided row (the stream vector ot.fv[s] in ascending order) */
static float idoutp_ms(const float *x, int D, const float *mean, const float *ivar, float gconst, const int *dimStream, int stream)
{
   int i;
   float sum, xmm;
   sum = gconst;
   for (i = 0; i < D; i++) {
      if (dimStream[i] != stream) continue;
      xmm = x[i] - mean[i];
      sum += xmm * xmm * ivar[i];
   }
   return -0.5 * sum;
}

/* ShStrP (HFB.c:898-988) for one stream element of a multi-stream state */
float orc_elem_outp(const orc_model *m, int e, const float *x, float *mixp_out)
{
   int c0 = m->stateCompOff[e], c1 = m->stateCompOff[e + 1], M = c1 - c0, k, g, str = e % m->NSt;
   float xx, mixp, wt, det = 0.0f;
   if (M == 1) {
      g = m->compGauss[c0];
      xx = idoutp_ms(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g], m->dimStream, str);
      xx += det;
      if (mixp_out) mixp_out[0] = xx;
      return xx;
   }
   xx = ORC_LZERO;
   for (k = 0; k < M; k++) {
      if (mixp_out) mixp_out[k] = ORC_LZERO;
      wt = m->compLogWt[c0 + k];
      if (wt > ORC_LMINMIX) {
         g = m->compGauss[c0 + k];
         mixp = idoutp_ms(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g], m->dimStream, str);
         mixp += det;
         xx = orc_ladd(xx, wt + mixp);
         if (mixp_out) mixp_out[k] = mixp;
      }
   }
   return xx;
}

/* HModel.c:5503-5555 SOutP, PLAINHS/SHAREDHS: LogDouble bx,px */
float orc_soutp(const orc_model *m, int s, const float *x)
{
   int c0 = m->stateCompOff[s], c1 = m->stateCompOff[s + 1], M = c1 - c0, k, g;
   double bx, px;
   float wt;
   if (M == 1) {
      g = m->compGauss[c0];
      return orc_idoutp(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g]);
   }
   bx = ORC_LZERO;
   for (k = 0; k < M; k++) {
      wt = m->compLogWt[c0 + k];
      if (wt > ORC_LMINMIX) {
         g = m->compGauss[c0 + k];
         px = orc_idoutp(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g]);
         bx = orc_ladd(bx, wt + px);
      }
   }
   return bx;
}

/* HModel.c:5347-5358 DOutP: the DIAGC form, xmm*xmm/var with the float division (sets that did not go through ConvDiagC) */
float orc_doutp(const float *x, int D, const float *mean, const float *var, float gconst)
{
   int i;
   float sum, xmm;
   sum = gconst;
   for (i = 0; i < D; i++) {
      xmm = x[i] - mean[i];
      sum += xmm * xmm / var[i];
   }
   return -0.5 * sum;
}

/* SOutP over a whole block in either covariance form: out[t*ns + k]; var == NULL takes the model's 1/variance (IDOutP) */
void orc_soutp_block(const orc_model *m, const float *var, const float *X, int T, const int *states, int ns, float *out)
{
   int t, k, c, g;
   for (t = 0; t < T; t++)
      for (k = 0; k < ns; k++) {
         const int s = states[k], c0 = m->stateCompOff[s], c1 = m->stateCompOff[s + 1];
         const float *x = X + (size_t)t * m->D;
         double bx = ORC_LZERO, px;
         for (c = c0; c < c1; c++) {
            const float wt = m->compLogWt[c];
            if (c1 - c0 > 1 && !(wt > ORC_LMINMIX)) continue;
            g = m->compGauss[c];
            px = var ? orc_doutp(x, m->D, m->mean + (size_t)g * m->D, var + (size_t)g * m->D, m->gconst[g])
                     : orc_idoutp(x, m->D, m->mean + (size_t)g * m->D, m->ivar + (size_t)g * m->D, m->gconst[g]);
            if (c1 - c0 == 1) bx = px; else bx = orc_ladd(bx, wt + px);
         }
         out[(size_t)t * ns + k] = bx;
      }
}

/* ShStrP's rounding with DOutP's covariance form (HTKAMD_SCORE_DIAGC alone) */
void orc_score_block_diagc(const orc_model *m, const float *var, const float *X, int T, const int *states, int ns, float *out)
{
   int t, k, c, g;
   for (t = 0; t < T; t++)
      for (k = 0; k < ns; k++) {
         const int s = states[k], c0 = m->stateCompOff[s], c1 = m->stateCompOff[s + 1];
         const float *x = X + (size_t)t * m->D;
         float xx = ORC_LZERO, mixp;
         for (c = c0; c < c1; c++) {
            const float wt = m->compLogWt[c];
            if (c1 - c0 > 1 && !(wt > ORC_LMINMIX)) continue;
            g = m->compGauss[c];
            mixp = orc_doutp(x, m->D, m->mean + (size_t)g * m->D, var + (size_t)g * m->D, m->gconst[g]);
            if (c1 - c0 == 1) xx = mixp; else xx = orc_ladd(xx, wt + mixp);
         }
         out[(size_t)t * ns + k] = xx;
      }
}

void orc_score_block(const orc_model *m, const float *X, int T, const int *states, int ns, float *out)
{
   int t, k;
   for (t = 0; t < T; t++)
      for (k = 0; k < ns; k++)
         out[(size_t)t * ns + k] = orc_state_outp(m, states[k], X + (size_t)t * m->D, NULL);
}

/* ------------------------------------------------------------------ forward-backward */

typedef struct {
   const orc_model *m;
   const orc_fbcfg *cfg;
   const float *X;
   int T, Q, maxN, maxM;
   const int *labs;
   int *N;            /* [Q+2] states of model q (1-based q) */
   const float **tp;  /* [Q+2] transition matrix of model q  */
   int *qDms;         /* [Q+2] */
   int *slotOff;      /* [Q+2] first emitting slot of model q */
   int nSlots;
   short *qLo, *qHi;  /* [T+2] */
   double pruneThresh;
   double *beta;      /* [(T+2)*(Q+2)*(maxN+1)] */
   unsigned char *bpres; /* [(T+2)*(Q+2)] beta[t][q] != NULL */
   float *outp;       /* [(T+1)*nSlots*(maxM+1)]: [0]=state prob, [1..M]=per-mixture */
   unsigned char *opres; /* [(T+1)*(Q+2)] otprob[t][q] computed */
   double *alphat, *alphat1; /* [(Q+2)*(maxN+1)] */
   float *occt;       /* [maxN+1] */
   long long nEval;
   /* several streams: otprob[t][q][j][s], s = 1..S -- pointers to the stream's probability vector ([0] = the stream's log
      probability, [1..M] per component), which a tied state met again at the same frame SHARES with its first visit (ShStrP's
      wa->prob, HFB.c:911-912) */
   int NSt;
   float **sv;        /* [(T+1)*nSlots*NSt] */
   int *lastT;        /* [S*NSt] wa->time */
   float **lastVec;   /* [S*NSt] wa->prob */
   float **blocks; int nBlocks, capBlocks; size_t blockUsed;
   /* tied mixtures: tmRecs[s] of the current frame -- probs (index, prob) sorted by prob, topM, maxP (HModel.h TMixRec) */
   int tmCap; int *tmIndex; float *tmProb; int *tmTopM; float *tmMaxP; int *tmOff;
} fbws;

#define SV_(t,q,j) (((size_t)(t) * w->nSlots + w->slotOff[q] + ((j) - 2)) * w->NSt)
#define ORC_ARENA 65536
static float *sv_alloc(fbws *w, int n)
{
   float *p;
   if (w->nBlocks == 0 || w->blockUsed + (size_t)n > ORC_ARENA) {
      if (w->nBlocks == w->capBlocks) { w->capBlocks = w->capBlocks * 2 + 16; w->blocks = (float **)realloc(w->blocks, sizeof(float *) * (size_t)w->capBlocks); }
      w->blocks[w->nBlocks++] = (float *)malloc(sizeof(float) * (n > ORC_ARENA ? (size_t)n : ORC_ARENA));
      w->blockUsed = 0;
   }
   p = w->blocks[w->nBlocks - 1] + w->blockUsed;
   w->blockUsed += (size_t)n;
   return p;
}

#define A_(q,i)   ((size_t)(q) * (w->maxN + 1) + (i))
#define B_(t,q,i) (((size_t)(t) * (w->Q + 2) + (q)) * (w->maxN + 1) + (i))
#define BP_(t,q)  ((size_t)(t) * (w->Q + 2) + (q))
#define O_(t,q,j) (((size_t)(t) * w->nSlots + w->slotOff[q] + ((j) - 2)) * (w->maxM + 1))
#define TPQ(q,i,j) (w->tp[q][((i)-1) * w->N[q] + ((j)-1)])

/* HFB.c:1116-1145 SetBeamTaper */
static void set_beam_taper(fbws *w)
{
   int q, dq, i, t, Q = w->Q, T = w->T;
   q = 1; dq = w->qDms[q]; i = 0;
   for (t = 1; t <= T; t++) {
      while (i == dq) {
         i = 0;
         if (q < Q) { q++; dq = w->qDms[q]; }
         else dq = -1;
      }
      w->qHi[t] = q;
      i++;
   }
   q = Q; dq = w->qDms[q]; i = 0;
   for (t = T; t >= 1; t--) {
      while (i == dq) {
         i = 0;
         if (q > 1) { q--; dq = w->qDms[q]; }
         else dq = -1;
      }
      w->qLo[t] = q;
      i++;
   }
}

typedef struct { int index; float prob; } tmprob;
static int cmp_tm(const void *a, const void *b)      /* CmpTM HModel.c:5298 */
{
   if (((const tmprob *)b)->prob < ((const tmprob *)a)->prob) return -1;
   if (((const tmprob *)b)->prob > ((const tmprob *)a)->prob) return +1;
   return 0;
}
/* PrecomputeTMix (HModel.c:5308-5346) with topM == 0: the pool's log probabilities at frame t, sorted, those within tmThresh of the
   best scaled by it and kept */
static void precompute_tmix(fbws *w, int t)
{
   const orc_model *m = w->m;
   const float *x = w->X + (size_t)(t - 1) * m->D;
   const float tmThresh = w->cfg->minFrwdP;
   int s, k;
   for (s = 0; s < w->NSt; s++) {
      const int c0 = m->stateCompOff[s], M = m->stateCompOff[s + 1] - c0;      /* the pool of stream s, as state 0 lists it */
      tmprob *pr = (tmprob *)malloc(sizeof(tmprob) * (size_t)M);
      float maxP = ORC_LZERO, minP, p;
      int mm;
      for (mm = 0; mm < M; mm++) {
         const int g = m->compGauss[c0 + mm];
         float sum = m->gconst[g], xmm;
         for (k = 0; k < m->D; k++) {             /* MOutP -> DOutP (HModel.c:5347) on the stream's vector */
            if (m->dimStream && m->dimStream[k] != s) continue;
            xmm = x[k] - m->mean[(size_t)g * m->D + k];
            sum += xmm * xmm / m->var[(size_t)g * m->D + k];
         }
         p = -0.5 * sum;
         if (p > maxP) maxP = p;
         pr[mm].prob = p; pr[mm].index = mm;
      }
      qsort(pr, (size_t)M, sizeof(tmprob), cmp_tm);
      minP = maxP - tmThresh;
      for (mm = 0; mm < M; mm++) {
         if (pr[mm].prob < minP) break;
         p = pr[mm].prob - maxP;
         pr[mm].prob = (p < ORC_MINEARG) ? 0.0 : exp(p);
      }
      w->tmTopM[s] = mm; w->tmMaxP[s] = maxP;
      for (mm = 0; mm < M; mm++) { w->tmIndex[w->tmOff[s] + mm] = pr[mm].index; w->tmProb[w->tmOff[s] + mm] = pr[mm].prob; }
      free(pr);
   }
}
/* SOutP, TIEDHS (HModel.c:5555-5566) for element e = state*NSt + stream */
static float tm_soutp(const fbws *w, int e)
{
   const orc_model *m = w->m;
   const int s = e % w->NSt, c0 = m->stateCompOff[e];
   double sum = 0.0;
   int mx;
   for (mx = 0; mx < w->tmTopM[s]; mx++)
      sum += w->tmProb[w->tmOff[s] + mx] * m->compWeight[c0 + w->tmIndex[w->tmOff[s] + mx]];
   return (sum >= ORC_MINLARG) ? log(sum) + w->tmMaxP[s] : ORC_LZERO;
}

/* HFB.c:991-1080 Setotprob for PLAINHS/SHAREDHS, S==1: evaluates models qLo-1(if >1)..qHi */
static void set_otprob(fbws *w, int t, int qHi, int qLo)
{
   int q, j;
   const orc_model *m = w->m;
   if (m->tiedMix) precompute_tmix(w, t);        /* HFB.c:1010 */
   if (qLo > 1) --qLo;
   for (q = qHi; q >= qLo; q--) {
      if (!w->opres[BP_(t, q)]) {
         int h = w->labs[q - 1];
         for (j = 2; j < w->N[q]; j++) {
            int s = m->hmmState[m->hmmStateOff[h] + (j - 2)];
            float *o = w->outp + O_(t, q, j);
            if (m->tiedMix) {                     /* HFB.c:1030-1040: SOutP per stream into fresh one-element vectors */
               if (w->NSt > 1) {
                  float **sv = w->sv + SV_(t, q, j), sum = 0.0;
                  int ks;
                  for (ks = 0; ks < w->NSt; ks++) { sv[ks] = sv_alloc(w, 1); sv[ks][0] = tm_soutp(w, s * w->NSt + ks); sum += sv[ks][0]; }
                  o[0] = sum;
                  for (ks = 0; ks < w->NSt; ks++) sv[ks][0] = sum - sv[ks][0];
               } else o[0] = tm_soutp(w, s);
               w->nEval++;
               continue;
            }
            if (w->NSt > 1) {                     /* HFB.c:1026-1066, PLAINHS / SHAREDHS with S > 1 */
               float **sv = w->sv + SV_(t, q, j), sum = 0.0;
               int ks, seenState = 0;
               for (ks = 0; ks < w->NSt; ks++) {
                  const int e = s * w->NSt + ks, M = m->stateCompOff[e + 1] - m->stateCompOff[e];
                  seenState = (!m->msIntended && w->lastT[e] == t);      /* :1044, overwritten stream by stream: the LAST stream's decides */
                  if (seenState) sv[ks] = w->lastVec[e];                  /* ShStrP :911-912 */
                  else {
                     float *v = sv_alloc(w, M == 1 ? 1 : M + 1);          /* NewOtprobVec :883 */
                     v[0] = orc_elem_outp(m, e, w->X + (size_t)(t - 1) * m->D, M == 1 ? NULL : v + 1);
                     w->lastT[e] = t; w->lastVec[e] = v;
                     sv[ks] = v;
                  }
                  sum += sv[ks][0];
               }
               if (seenState) o[0] = sum / 2;                             /* :1059 */
               else {
                  o[0] = sum;
                  for (ks = 0; ks < w->NSt; ks++) sv[ks][0] = sum - sv[ks][0];      /* :1062-1064: in the SHARED vector */
               }
               w->nEval++;
               continue;
            }
            o[0] = orc_state_outp(m, s, w->X + (size_t)(t - 1) * m->D, o + 1);
            w->nEval++;
         }
         w->opres[BP_(t, q)] = 1;
      }
   }
}

/* HFB.c:1149-1296 SetBeta.  Returns utt pr or LZERO. */
static double set_beta(fbws *w)
{
   int i, j, t, q, Nq, lNq = 0, startq, endq;
   int Q = w->Q, T = w->T;
   double x, y, gMax, lMax, a, a1N = 0.0;
   double *bqt = NULL, *bqt1, *bq1t1;
   double *maxP = (double *)malloc(sizeof(double) * (Q + 2));
   double pr;

   /* Last Column t = T */
   w->qHi[T] = Q; endq = w->qLo[T];
   set_otprob(w, T, Q, endq);
   gMax = ORC_LZERO;
   for (q = Q; q >= endq; q--) {
      Nq = w->N[q];
      bqt = w->beta + B_(T, q, 0); w->bpres[BP_(T, q)] = 1;
      bqt[Nq] = (q == Q) ? 0.0 : w->beta[B_(T, q + 1, lNq)] + a1N;
      for (i = 2; i < Nq; i++)
         bqt[i] = TPQ(q, i, Nq) + bqt[Nq];
      x = ORC_LZERO;
      for (j = 2; j < Nq; j++) {
         a = TPQ(q, 1, j); y = bqt[j];
         if (a > ORC_LSMALL && y > ORC_LSMALL)
            x = orc_ladd(x, a + w->outp[O_(T, q, j)] + y);
      }
      bqt[1] = x;
      lNq = Nq; a1N = TPQ(q, 1, Nq);
      if (x > gMax) gMax = x;
   }

   /* Columns T-1 -> 1 */
   for (t = T - 1; t >= 1; t--) {
      gMax = ORC_LZERO;
      startq = w->qHi[t + 1];
      endq = (w->qLo[t + 1] == 1) ? 1 : ((w->qLo[t] >= w->qLo[t + 1]) ? w->qLo[t] : w->qLo[t + 1] - 1);
      while (endq > 1 && w->qDms[endq - 1] == 0) endq--;
      set_otprob(w, t, startq, endq);
      for (q = startq; q >= endq; q--) {
         lMax = ORC_LZERO;
         Nq = w->N[q];
         bqt = w->beta + B_(t, q, 0); w->bpres[BP_(t, q)] = 1;
         bqt1 = w->bpres[BP_(t + 1, q)] ? w->beta + B_(t + 1, q, 0) : NULL;
         bq1t1 = (q == Q) ? NULL : (w->bpres[BP_(t + 1, q + 1)] ? w->beta + B_(t + 1, q + 1, 0) : NULL);
         bqt[Nq] = (bq1t1 == NULL) ? ORC_LZERO : bq1t1[1];
         if (q < startq && a1N > ORC_LSMALL)
            bqt[Nq] = orc_ladd(bqt[Nq], w->beta[B_(t, q + 1, lNq)] + a1N);
         for (i = Nq - 1; i > 1; i--) {
            x = TPQ(q, i, Nq) + bqt[Nq];
            if (q >= w->qLo[t + 1] && q <= w->qHi[t + 1])
               for (j = 2; j < Nq; j++) {
                  a = TPQ(q, i, j); y = bqt1[j];
                  if (a > ORC_LSMALL && y > ORC_LSMALL)
                     x = orc_ladd(x, a + w->outp[O_(t + 1, q, j)] + y);
               }
            bqt[i] = x;
            if (x > lMax) lMax = x;
            if (x > gMax) gMax = x;
         }
         x = ORC_LZERO;
         for (j = 2; j < Nq; j++) {
            a = TPQ(q, 1, j);
            y = bqt[j];
            if (a > ORC_LSMALL && y > ORC_LSMALL)
               x = orc_ladd(x, a + w->outp[O_(t, q, j)] + y);
         }
         bqt[1] = x;
         maxP[q] = lMax;
         lNq = Nq; a1N = TPQ(q, 1, Nq);
      }
      while (gMax - maxP[startq] > w->pruneThresh) {
         w->bpres[BP_(t, startq)] = 0;
         --startq;
         if (startq < 1) { free(maxP); return ORC_LZERO; }  /* HError 7323 in the reference */
      }
      while (w->qHi[t] < startq) {
         w->bpres[BP_(t, startq)] = 0;
         --startq;
         if (startq < 1) { free(maxP); return ORC_LZERO; }
      }
      w->qHi[t] = startq;
      while (gMax - maxP[endq] > w->pruneThresh) {
         w->bpres[BP_(t, endq)] = 0;
         ++endq;
         if (endq > startq) { free(maxP); return ORC_LZERO; }
      }
      w->qLo[t] = endq;
   }
   pr = bqt[1];   /* bqt is beta[1][qLo-at-t=1 loop end]: last q processed == endq of t=1, see note */
   free(maxP);
   if (pr <= ORC_LSMALL) return ORC_LZERO;
   return pr;
}

/* HFB.c:600-613 ZeroAlpha */
static void zero_alpha(fbws *w, int qlo, int qhi)
{
   int q, j;
   for (q = qlo; q <= qhi; q++)
      for (j = 1; j <= w->N[q]; j++)
         w->alphat[A_(q, j)] = ORC_LZERO;
}

/* HFB.c:616-651 InitAlpha */
static void init_alpha(fbws *w, int *start, int *end)
{
   int i, j, Nq, eq, q;
   double x, a, a1N = 0.0;
   eq = w->qHi[1];
   for (q = 1; q <= eq; q++) {
      double *aq = w->alphat + A_(q, 0);
      Nq = w->N[q];
      aq[1] = (q == 1) ? 0.0 : w->alphat[A_(q - 1, 1)] + a1N;
      for (j = 2; j < Nq; j++) {
         a = TPQ(q, 1, j);
         aq[j] = (a > ORC_LSMALL) ? aq[1] + a + w->outp[O_(1, q, j)] : ORC_LZERO;
      }
      x = ORC_LZERO;
      for (i = 2; i < Nq; i++) {
         a = TPQ(q, i, Nq);
         if (a > ORC_LSMALL)
            x = orc_ladd(x, aq[i] + a);
      }
      aq[Nq] = x;
      a1N = TPQ(q, 1, Nq);
   }
   zero_alpha(w, eq + 1, w->Q);
   *start = 1; *end = eq;
}

/* HFB.c:655-682 MaxModelProb */
static double max_model_prob(fbws *w, int q, int t, int minq)
{
   double maxP, x;
   int Nq1, Nq, i, qx, qx1;
   if (q == 1)
      maxP = ORC_LZERO;
   else {
      Nq1 = w->N[q - 1];
      maxP = (!w->bpres[BP_(t, q - 1)]) ? ORC_LZERO : w->alphat[A_(q - 1, Nq1)] + w->beta[B_(t, q - 1, Nq1)];
      for (qx = q - 1; qx > minq && TPQ(qx, 1, Nq1) > ORC_LSMALL; qx--) {
         qx1 = qx - 1;
         Nq1 = w->N[qx1];
         x = (!w->bpres[BP_(t, qx1)]) ? ORC_LZERO : w->alphat[A_(qx1, Nq1)] + w->beta[B_(t, qx1, Nq1)];
         if (x > maxP) maxP = x;
      }
   }
   Nq = w->N[q];
   if (w->bpres[BP_(t, q)]) {
      for (i = 1; i < Nq; i++)
         if ((x = w->alphat[A_(q, i)] + w->beta[B_(t, q, i)]) > maxP) maxP = x;
   }
   return maxP;
}

/* HFB.c:686-784 StepAlpha; returns 0 or the reference's fatal error code */
static int step_alpha(fbws *w, int t, int *start, int *end, double pr)
{
   int sq, eq, i, j, q, Nq, lNq, Q = w->Q;
   double x = 0.0, y, a, a1N = 0.0, *tmp;

   sq = w->qLo[t - 1];
   while (pr - max_model_prob(w, sq, t - 1, sq) > w->cfg->minFrwdP) {
      ++sq;
      if (sq > w->qHi[t]) return -7390;
   }
   if (sq < w->qLo[t]) sq = w->qLo[t];

   eq = w->qHi[t - 1] < Q ? w->qHi[t - 1] + 1 : w->qHi[t - 1];
   while (pr - max_model_prob(w, eq, t - 1, sq) > w->cfg->minFrwdP) {
      --eq;
      if (eq < sq) return -7390;
   }
   while (eq < Q && w->qDms[eq] == 0) eq++;
   if (eq > w->qHi[t]) eq = w->qHi[t];

   tmp = w->alphat1; w->alphat1 = w->alphat; w->alphat = tmp;

   if (sq > 1) zero_alpha(w, 1, sq - 1);
   Nq = (sq == 1) ? 0 : w->N[sq - 1];

   for (q = sq; q <= eq; q++) {
      double *aq, *laq;
      lNq = Nq; Nq = w->N[q];
      aq = w->alphat + A_(q, 0);
      laq = w->alphat1 + A_(q, 0);
      if (q == 1)
         aq[1] = ORC_LZERO;
      else {
         aq[1] = w->alphat1[A_(q - 1, lNq)];
         if (q > sq && a1N > ORC_LSMALL)
            aq[1] = orc_ladd(aq[1], w->alphat[A_(q - 1, 1)] + a1N);
      }
      for (j = 2; j < Nq; j++) {
         a = TPQ(q, 1, j);
         x = (a > ORC_LSMALL) ? a + aq[1] : ORC_LZERO;
         for (i = 2; i < Nq; i++) {
            a = TPQ(q, i, j); y = laq[i];
            if (a > ORC_LSMALL && y > ORC_LSMALL)
               x = orc_ladd(x, y + a);
         }
         aq[j] = x + w->outp[O_(t, q, j)];
      }
      x = ORC_LZERO;
      for (i = 2; i < Nq; i++) {
         a = TPQ(q, i, Nq); y = aq[i];
         if (a > ORC_LSMALL && y > ORC_LSMALL)
            x = orc_ladd(x, y + a);
      }
      aq[Nq] = x; a1N = TPQ(q, 1, Nq);
   }
   if (eq < Q) zero_alpha(w, eq + 1, Q);
   *start = sq; *end = eq;
   return 0;
}

/* HFB.c:399-418 SetOcct */
static void set_occt(fbws *w, int q, const double *aqt, const double *bqt, const double *bq1t, double pr)
{
   int i, N = w->N[q];
   double x;
   for (i = 1; i <= N; i++) {
      x = aqt[i] + bqt[i];
      if (i == 1 && bq1t != NULL && TPQ(q, 1, N) > ORC_LSMALL)
         x = orc_ladd(x, aqt[1] + bq1t[1] + TPQ(q, 1, N));
      x -= pr;
      w->occt[i] = (x > ORC_MINEARG) ? exp(x) : 0.0;
   }
}

/* HFB.c:1371-1423 UpTranParms */
static void up_tran_parms(fbws *w, orc_accs *acc, int t, int q, const double *aqt, const double *bqt,
                          const double *bqt1, const double *bq1t, double pr)
{
   const orc_model *m = w->m;
   int i, j, N = w->N[q], ti, k, occOff = 0;
   float *tran, *occ;
   double x;
   ti = m->hmmTrans[w->labs[q - 1]];
   for (k = 0; k < ti; k++) occOff += m->transN[k];
   tran = acc->tr + m->transOff[ti];
   occ = acc->trOcc + occOff;
   for (i = 1; i < N; i++)
      occ[i - 1] += w->occt[i];
   for (i = 1; i < N; i++) {
      float *trow = tran + (size_t)(i - 1) * N;   /* ti[j] -> trow[j-1] */
      for (j = 2; j <= N; j++) {
         if (i == 1 && j < N) {
            x = aqt[1] + TPQ(q, 1, j) + w->outp[O_(t, q, j)] + bqt[j] - pr;
            if (x > ORC_MINEARG) trow[j - 1] += exp(x);
         } else if (i > 1 && j < N && bqt1 != NULL) {
            x = aqt[i] + TPQ(q, i, j) + w->outp[O_(t + 1, q, j)] + bqt1[j] - pr;
            if (x > ORC_MINEARG) trow[j - 1] += exp(x);
         } else if (i > 1 && j == N) {
            x = aqt[i] + TPQ(q, i, N) + bqt[N] - pr;
            if (x > ORC_MINEARG) trow[N - 1] += exp(x);
         }
         if (i == 1 && j == N && TPQ(q, 1, N) > ORC_LSMALL && bq1t != NULL) {
            x = aqt[1] + TPQ(q, 1, N) + bq1t[1] - pr;
            if (x > ORC_MINEARG) trow[N - 1] += exp(x);
         }
      }
   }
}

/* HFB.c:1426-1744 UpMixParms, PLAINHS/SHAREDHS, S==1, single model set, DIAGC/INVDIAGC, no xforms */
static void up_mix_parms(fbws *w, orc_accs *acc, int t, int q, const double *aqt, const double *aqt1,
                         const double *bqt, double pr)
{
   const orc_model *m = w->m;
   int i, j, k, mx, M, N = w->N[q], D = m->D, uF = w->cfg->uFlags;
   int h = w->labs[q - 1];
   float a, c_jm, prob, wght, zmean, zmeanlr;
   double x, initx = ORC_LZERO, Lr, steSumLr;

   for (j = 2; j < N; j++) {
      int s = m->hmmState[m->hmmStateOff[h] + (j - 2)];
      int c0 = m->stateCompOff[w->NSt > 1 ? s * w->NSt : s];
      const float *outprob = w->outp + O_(t, q, j);
      const float *ot = w->X + (size_t)(t - 1) * D;
      if (w->maxM > 1) {                          /* fbInfo->maxM: maximum over the whole set */
         initx = TPQ(q, 1, j) + aqt[1];
         if (t > 1)
            for (i = 2; i < N; i++) {
               a = TPQ(q, i, j);
               if (a > ORC_LSMALL)
                  initx = orc_ladd(initx, aqt1[i] + a);
            }
         initx += bqt[j] - pr;
      }
      if (m->tiedMix) {                           /* UpMixParms, TIEDHS: the non-pruned components of the pool (HFB.c:1503-1507,1559-1563,1597-1600) */
         int ks;
         for (ks = 0; ks < w->NSt; ks++) {
            const int e = s * w->NSt + ks;
            const float others = (w->NSt > 1) ? w->sv[SV_(t, q, j) + ks][0] : 0.0f;
            c0 = m->stateCompOff[e];
            steSumLr = 0.0;
            for (mx = 0; mx < w->tmTopM[ks]; mx++) {
               const int mi = w->tmIndex[w->tmOff[ks] + mx], c = c0 + mi, g = m->compGauss[c];
               float tmp;
               wght = orc_mix_log_weight(m->compWeight[c]);
               if (wght > ORC_LMINMIX) {
                  c_jm = wght;
                  x = initx + c_jm;
                  tmp = w->tmProb[w->tmOff[ks] + mx];
                  prob = (tmp >= ORC_MINLARG) ? log(tmp) + w->tmMaxP[ks] : ORC_LZERO;
                  x += prob;
                  if (w->NSt > 1) x += others;
                  if (-x < w->cfg->minFrwdP) {
                     const float *mean = m->mean + (size_t)g * D;
                     Lr = exp(x);
                     steSumLr += Lr;
                     if ((uF & ORC_UPMEANS) && (uF & ORC_UPVARS)) {
                        float *mu_jm = acc->mu + (size_t)g * D, *var = acc->va + (size_t)g * D;
                        acc->muOcc[g] += Lr;
                        acc->vaOcc[g] += Lr;
                        for (k = 0; k < D; k++) {
                           if (m->dimStream && m->dimStream[k] != ks) continue;
                           zmean = ot[k] - mean[k];
                           zmeanlr = zmean * Lr;
                           mu_jm[k] += zmeanlr;
                           var[k] += zmean * zmeanlr;
                        }
                     } else if (uF & ORC_UPMEANS) {
                        float *mu_jm = acc->mu + (size_t)g * D;
                        acc->muOcc[g] += Lr;
                        for (k = 0; k < D; k++) if (!m->dimStream || m->dimStream[k] == ks) mu_jm[k] += (ot[k] - mean[k]) * Lr;
                     } else if (uF & ORC_UPVARS) {
                        float *var = acc->va + (size_t)g * D;
                        acc->vaOcc[g] += Lr;
                        for (k = 0; k < D; k++) {
                           if (m->dimStream && m->dimStream[k] != ks) continue;
                           zmean = ot[k] - mean[k];
                           var[k] += zmean * zmean * Lr;
                        }
                     }
                     if (uF & ORC_UPMIXES)
                        acc->wt[c] += Lr;
                  }
               }
            }
            acc->wtOcc[e] += steSumLr;
         }
         continue;
      }
      if (w->NSt > 1) {                           /* the stream loop of UpMixParms (HFB.c:1499-1721) */
         int ks;
         for (ks = 0; ks < w->NSt; ks++) {
            const int e = s * w->NSt + ks;
            const float *op = w->sv[SV_(t, q, j) + ks];
            c0 = m->stateCompOff[e];
            M = m->stateCompOff[e + 1] - c0;
            steSumLr = 0.0;
            for (mx = 1; mx <= M; mx++) {
               int c = c0 + mx - 1, g = m->compGauss[c];
               wght = m->compLogWt[c];
               if (wght > ORC_LMINMIX) {
                  if (M == 1) x = aqt[j] + bqt[j] - pr;
                  else {
                     c_jm = wght;
                     x = initx + c_jm;
                     prob = op[mx];
                     x += prob;
                     x += op[0];                  /* adjust for parallel streams :1611 */
                  }
                  if (-x < w->cfg->minFrwdP) {
                     const float *mean = m->mean + (size_t)g * D;
                     Lr = exp(x);
                     steSumLr += Lr;
                     if ((uF & ORC_UPMEANS) && (uF & ORC_UPVARS)) {
                        float *mu_jm = acc->mu + (size_t)g * D, *var = acc->va + (size_t)g * D;
                        acc->muOcc[g] += Lr;
                        acc->vaOcc[g] += Lr;
                        for (k = 0; k < D; k++) {
                           if (m->dimStream[k] != ks) continue;
                           zmean = ot[k] - mean[k];
                           zmeanlr = zmean * Lr;
                           mu_jm[k] += zmeanlr;
                           var[k] += zmean * zmeanlr;
                        }
                     } else if (uF & ORC_UPMEANS) {
                        float *mu_jm = acc->mu + (size_t)g * D;
                        acc->muOcc[g] += Lr;
                        for (k = 0; k < D; k++) if (m->dimStream[k] == ks) mu_jm[k] += (ot[k] - mean[k]) * Lr;
                     } else if (uF & ORC_UPVARS) {
                        float *var = acc->va + (size_t)g * D;
                        acc->vaOcc[g] += Lr;
                        for (k = 0; k < D; k++) {
                           if (m->dimStream[k] != ks) continue;
                           zmean = ot[k] - mean[k];
                           var[k] += zmean * zmean * Lr;
                        }
                     }
                     if (uF & ORC_UPMIXES)
                        acc->wt[c] += Lr;
                  }
               }
            }
            acc->wtOcc[e] += steSumLr;
         }
         continue;
      }
      M = m->stateCompOff[s + 1] - c0;
      steSumLr = 0.0;
      for (mx = 1; mx <= M; mx++) {
         int c = c0 + mx - 1, g = m->compGauss[c];
         wght = m->compLogWt[c];
         if (wght > ORC_LMINMIX) {
            if (M == 1)                            /* !mmix */
               x = aqt[j] + bqt[j] - pr;
            else {
               c_jm = wght;
               x = initx + c_jm;
               prob = outprob[mx];
               x += prob;
            }
            if (-x < w->cfg->minFrwdP) {
               const float *mean = m->mean + (size_t)g * D;
               Lr = exp(x);
               steSumLr += Lr;
               if ((uF & ORC_UPMEANS) && (uF & ORC_UPVARS)) {
                  float *mu_jm = acc->mu + (size_t)g * D, *var = acc->va + (size_t)g * D;
                  acc->muOcc[g] += Lr;
                  acc->vaOcc[g] += Lr;
                  for (k = 0; k < D; k++) {
                     zmean = ot[k] - mean[k];
                     zmeanlr = zmean * Lr;
                     mu_jm[k] += zmeanlr;
                     var[k] += zmean * zmeanlr;
                  }
               } else if (uF & ORC_UPMEANS) {
                  float *mu_jm = acc->mu + (size_t)g * D;
                  acc->muOcc[g] += Lr;
                  for (k = 0; k < D; k++)
                     mu_jm[k] += (ot[k] - mean[k]) * Lr;
               } else if (uF & ORC_UPVARS) {
                  float *var = acc->va + (size_t)g * D;
                  acc->vaOcc[g] += Lr;
                  for (k = 0; k < D; k++) {
                     zmean = ot[k] - mean[k];
                     var[k] += zmean * zmean * Lr;
                  }
               }
               if (uF & ORC_UPMIXES)
                  acc->wt[c] += Lr;
            }
         }
      }
      acc->wtOcc[s] += steSumLr;
   }
}

int orc_fb_utt(const orc_model *m, const orc_fbcfg *cfg, const float *X, int T,
               const int *labs, int Q, orc_accs *acc, double *pr_out, orc_fbdump *dump)
{
   fbws ws, *w = &ws;
   int q, t, i, qt, start = 0, end = 0, rc = 1, h;
   double lbeta = ORC_LZERO, pr;
   size_t nb;

   memset(w, 0, sizeof(*w));
   w->m = m; w->cfg = cfg; w->X = X; w->T = T; w->Q = Q; w->labs = labs;
   w->N = (int *)calloc(Q + 2, sizeof(int));
   w->tp = (const float **)calloc(Q + 2, sizeof(float *));
   w->qDms = (int *)calloc(Q + 2, sizeof(int));
   w->slotOff = (int *)calloc(Q + 2, sizeof(int));
   w->qLo = (short *)calloc(T + 2, sizeof(short));
   w->qHi = (short *)calloc(T + 2, sizeof(short));
   w->maxM = 1;
   w->NSt = m->NSt > 1 ? m->NSt : 1;
   for (i = 0; i < m->S * w->NSt; i++) {  /* MaxMixInSet (HFB.c:261) is a property of the whole set */
      int M = m->stateCompOff[i + 1] - m->stateCompOff[i];
      if (M > w->maxM) w->maxM = M;
   }
   /* CreateInsts HFB.c:508-574 */
   qt = 0; w->maxN = 0; w->nSlots = 0;
   for (q = 1; q <= Q; q++) {
      int ti;
      h = labs[q - 1];
      ti = m->hmmTrans[h];
      w->N[q] = m->transN[ti];
      w->tp[q] = m->transP + m->transOff[ti];
      w->qDms[q] = orc_min_dur(w->N[q], w->tp[q]);
      qt += w->qDms[q];
      if (w->N[q] > w->maxN) w->maxN = w->N[q];
      w->slotOff[q] = w->nSlots;
      w->nSlots += w->N[q] - 2;
      if (q > 1 && w->qDms[q] == 0 && w->qDms[q - 1] == 0) { rc = -7332; goto done0; }
   }
   if (w->qDms[1] == 0 || w->qDms[Q] == 0) { rc = -7332; goto done0; }
   if (qt > T) { rc = 0; goto done0; }     /* HFB.c:1339-1344 */

   nb = (size_t)(T + 2) * (Q + 2) * (w->maxN + 1);
   w->beta = (double *)malloc(nb * sizeof(double));
   w->bpres = (unsigned char *)malloc((size_t)(T + 2) * (Q + 2));
   w->outp = (float *)malloc((size_t)(T + 1) * w->nSlots * (w->maxM + 1) * sizeof(float));
   w->opres = (unsigned char *)malloc((size_t)(T + 2) * (Q + 2));
   w->alphat = (double *)malloc((size_t)(Q + 2) * (w->maxN + 1) * sizeof(double));
   w->alphat1 = (double *)malloc((size_t)(Q + 2) * (w->maxN + 1) * sizeof(double));
   w->occt = (float *)malloc((w->maxN + 1) * sizeof(float));
   if (m->tiedMix) {
      int tot = 0, ks;
      w->tmOff = (int *)calloc((size_t)w->NSt + 1, sizeof(int));
      for (ks = 0; ks < w->NSt; ks++) { w->tmOff[ks] = tot; tot += m->stateCompOff[ks + 1] - m->stateCompOff[ks]; }
      w->tmIndex = (int *)malloc(sizeof(int) * (size_t)tot); w->tmProb = (float *)malloc(sizeof(float) * (size_t)tot);
      w->tmTopM = (int *)calloc((size_t)w->NSt, sizeof(int)); w->tmMaxP = (float *)calloc((size_t)w->NSt, sizeof(float));
   }
   if (w->NSt > 1) {
      w->sv = (float **)calloc((size_t)(T + 1) * w->nSlots * w->NSt, sizeof(float *));
      w->lastT = (int *)malloc(sizeof(int) * (size_t)m->S * w->NSt);
      w->lastVec = (float **)calloc((size_t)m->S * w->NSt, sizeof(float *));
   }

   /* StepBack HFB.c:1321-1366 */
   w->pruneThresh = cfg->pruneInit;
   for (;;) {
      memset(w->bpres, 0, (size_t)(T + 2) * (Q + 2));
      memset(w->opres, 0, (size_t)(T + 2) * (Q + 2));
      if (w->NSt > 1) for (i = 0; i < m->S * w->NSt; i++) w->lastT[i] = -1;     /* a new pass re-creates otprob; ResetHMMPreComps / ResetHMMWtAccs (HFB.c:559-562) */
      set_beam_taper(w);
      lbeta = set_beta(w);
      if (lbeta > ORC_LSMALL) break;
      w->pruneThresh += cfg->pruneInc;
      if (w->pruneThresh > cfg->pruneLim || cfg->pruneInc == 0.0) { rc = 0; goto done; }
   }
   pr = lbeta;
   *pr_out = pr;

   if (dump) {
      size_t n = (size_t)T * Q * w->maxN, k;
      if (dump->beta) for (k = 0; k < n; k++) dump->beta[k] = NAN;
      if (dump->alpha) for (k = 0; k < n; k++) dump->alpha[k] = NAN;
      if (dump->outp) for (k = 0; k < n; k++) dump->outp[k] = NAN;
      if (dump->occ) for (k = 0; k < n; k++) dump->occ[k] = NAN;
      for (t = 1; t <= T; t++) {
         if (dump->qLo) dump->qLo[t - 1] = w->qLo[t];
         if (dump->qHi) dump->qHi[t - 1] = w->qHi[t];
         for (q = 1; q <= Q; q++) {
            if (dump->beta && w->bpres[BP_(t, q)])
               for (i = 1; i <= w->N[q]; i++)
                  dump->beta[((size_t)(t - 1) * Q + (q - 1)) * w->maxN + (i - 1)] = w->beta[B_(t, q, i)];
            if (dump->outp && w->opres[BP_(t, q)])
               for (i = 2; i < w->N[q]; i++)
                  dump->outp[((size_t)(t - 1) * Q + (q - 1)) * w->maxN + (i - 1)] = w->outp[O_(t, q, i)];
         }
      }
      dump->nEval = w->nEval;
   }

   /* StepForward HFB.c:1752-1810 */
   init_alpha(w, &start, &end);
   for (q = 1; q <= Q; q++)
      acc->nEgs[labs[q - 1]] += 1;
   for (t = 1; t <= T; t++) {
      if (m->tiedMix) precompute_tmix(w, t);     /* StepForward HFB.c:1780 */
      if (t > 1) {
         int e = step_alpha(w, t, &start, &end, pr);
         if (e != 0) { rc = e; goto done; }
      }
      if (dump) {
         if (dump->aLo) dump->aLo[t - 1] = start;
         if (dump->aHi) dump->aHi[t - 1] = end;
         if (dump->alpha)
            for (q = 1; q <= Q; q++)
               for (i = 1; i <= w->N[q]; i++)
                  dump->alpha[((size_t)(t - 1) * Q + (q - 1)) * w->maxN + (i - 1)] = w->alphat[A_(q, i)];
      }
      for (q = start; q <= end; q++) {
         const double *aqt = w->alphat + A_(q, 0);
         const double *bqt = w->beta + B_(t, q, 0);
         const double *bqt1 = (t == T) ? NULL : (w->bpres[BP_(t + 1, q)] ? w->beta + B_(t + 1, q, 0) : NULL);
         const double *aqt1 = (t == 1) ? NULL : w->alphat1 + A_(q, 0);
         const double *bq1t = (q == Q) ? NULL : (w->bpres[BP_(t, q + 1)] ? w->beta + B_(t, q + 1, 0) : NULL);
         set_occt(w, q, aqt, bqt, bq1t, pr);
         if (dump && dump->occ)
            for (i = 1; i <= w->N[q]; i++)
               dump->occ[((size_t)(t - 1) * Q + (q - 1)) * w->maxN + (i - 1)] = w->occt[i];
         if (cfg->uFlags & (ORC_UPMEANS | ORC_UPVARS | ORC_UPMIXES))
            up_mix_parms(w, acc, t, q, aqt, aqt1, bqt, pr);
         if (cfg->uFlags & ORC_UPTRANS)
            up_tran_parms(w, acc, t, q, aqt, bqt, bqt1, bq1t, pr);
      }
   }
done:
   free(w->beta); free(w->bpres); free(w->outp); free(w->opres);
   free(w->alphat); free(w->alphat1); free(w->occt);
   free(w->sv); free(w->lastT); free(w->lastVec);
   free(w->tmOff); free(w->tmIndex); free(w->tmProb); free(w->tmTopM); free(w->tmMaxP);
   for (i = 0; i < w->nBlocks; i++) free(w->blocks[i]);
   free(w->blocks);
done0:
   free(w->N); free((void *)w->tp); free(w->qDms); free(w->slotOff); free(w->qLo); free(w->qHi);
   return rc;
}

/* ------------------------------------------------------------------ model update */

/* HERest.c:1262-1321 MLUpdateModels for PLAINHS/SHAREDHS, DIAGC, S==1.
   Shared structures are updated by the first physical HMM (scan order) that has >= minEgs examples;
   the outcome does not depend on which one, so physical index order is used here. */
void orc_update(const orc_model *m, const orc_accs *acc, const orc_updcfg *cfg,
                float *mean, float *var, float *gconst, float *compWeight, float *transP,
                orc_updstats *st)
{
   int h, i, j, k, c, D = m->D, maxM = 1;
   unsigned char *doneT = (unsigned char *)calloc(m->nT, 1);
   unsigned char *doneS = (unsigned char *)calloc(m->S, 1);
   unsigned char *doneMu = (unsigned char *)calloc(m->G, 1);
   unsigned char *doneVa = (unsigned char *)calloc(m->G, 1);
   int *occOff = (int *)calloc(m->nT + 1, sizeof(int));
   memset(st, 0, sizeof(*st));
   for (i = 0; i < m->nT; i++) occOff[i + 1] = occOff[i] + m->transN[i];
   for (i = 0; i < m->S; i++) {
      int M = m->stateCompOff[i + 1] - m->stateCompOff[i];
      if (M > maxM) maxM = M;
   }
   if (cfg->singleProcess) {
      /* HERest.c:1336-1339: the set is INVDIAGC with log weights at this point; ForceDiagC (HUtil.c:441)
         inverts the inverse variances again and ConvExpWt (HUtil.c:488) exponentiates the float log weights,
         so parameters that are NOT re-estimated come back through a float round trip. */
      /* ConvDiagC / ForceDiagC / ConvLogWt / ConvExpWt walk the set with an HMM scan: a state macro that no model uses is not
         visited and keeps its values (and, in a file written afterwards, has no <GCONST>) */
      unsigned char *usedS = (unsigned char *)calloc((size_t)m->S, 1), *usedG = (unsigned char *)calloc((size_t)m->G, 1);
      size_t z;
      int g;
      for (h = 0; h < m->H; h++)
         for (j = m->hmmStateOff[h]; j < m->hmmStateOff[h + 1]; j++) usedS[m->hmmState[j]] = 1;
      for (i = 0; i < m->S; i++)
         if (usedS[i]) for (c = m->stateCompOff[i]; c < m->stateCompOff[i + 1]; c++) usedG[m->compGauss[c]] = 1;
      for (g = 0; g < m->G; g++) {
         if (!usedG[g]) continue;
         for (z = (size_t)g * D; z < (size_t)(g + 1) * D; z++) {
            float v = var[z], iv;
            if (v > 1E+30) v = 1E+30;
            if (v < 1E-30) v = 1E-30;
            iv = 1 / v;
            if (iv > 1E+30) iv = 1E+30;
            if (iv < 1E-30) iv = 1E-30;
            var[z] = 1 / iv;
         }
      }
      for (i = 0; i < m->S; i++) {
         if (!usedS[i]) continue;
         for (c = m->stateCompOff[i]; c < m->stateCompOff[i + 1]; c++) {
            float lw = orc_mix_log_weight(compWeight[c]);
            compWeight[c] = exp(lw);
         }
      }
      free(usedS); free(usedG);
   }
   for (h = 0; h < m->H; h++) {
      int n = acc->nEgs[h], ti = m->hmmTrans[h], N = m->transN[ti];
      if (n < cfg->minEgs) st->nSkippedHmm++;
      if (!(n >= cfg->minEgs && n > 0)) continue;
      /* UpdateTrans HERest.c:795-816 */
      if ((cfg->uFlags & ORC_UPTRANS) && !doneT[ti]) {
         float *tp = transP + m->transOff[ti];
         const float *tran = acc->tr + m->transOff[ti], *occ = acc->trOcc + occOff[ti];
         for (i = 1; i < N; i++) {
            float occi = occ[i - 1], x;
            if (occi > 0.0)
               for (j = 2; j <= N; j++) {
                  x = tran[(i - 1) * N + (j - 1)] / occi;
                  tp[(i - 1) * N + (j - 1)] = (x > ORC_MINLARG) ? log(x) : ORC_LZERO;
               }
         }
         doneT[ti] = 1;
      }
      /* UpdateWeights HERest.c:897-971 */
      if (maxM > 1 && (cfg->uFlags & ORC_UPMIXES))
         for (j = 2; j < N; j++) {
            int s = m->hmmState[m->hmmStateOff[h] + (j - 2)];
            int c0 = m->stateCompOff[s], M = m->stateCompOff[s + 1] - c0;
            float occi = acc->wtOcc[s], x;
            if (doneS[s]) continue;
            if (occi > 0) {
               for (k = 0; k < M; k++) {
                  x = acc->wt[c0 + k] / occi;
                  if (x > 1.0) x = 1.0;                    /* >1.001 is HError 2393 in the reference */
                  compWeight[c0 + k] = (x > ORC_MINMIX) ? x : 0.0;
               }
               if (cfg->mixWeightFloor > 0.0) {           /* FloorMixes HERest.c:819-840 */
                  float sum = 0.0, fsum = 0.0, scale, floor = cfg->mixWeightFloor;
                  for (k = 0; k < M; k++) {
                     if (compWeight[c0 + k] > floor) sum += compWeight[c0 + k];
                     else { fsum += floor; compWeight[c0 + k] = floor; }
                  }
                  if (fsum != 0.0 && sum != 0.0) {
                     scale = (1.0 - fsum) / sum;
                     for (k = 0; k < M; k++)
                        if (compWeight[c0 + k] > floor) compWeight[c0 + k] *= scale;
                  }
               }
            }
            doneS[s] = 1;
         }
      /* UpdateVars HERest.c:1045-1122 (all states, BEFORE the means) */
      if (cfg->uFlags & ORC_UPVARS)
         for (j = 2; j < N; j++) {
            int s = m->hmmState[m->hmmStateOff[h] + (j - 2)];
            for (c = m->stateCompOff[s]; c < m->stateCompOff[s + 1]; c++)
               if (compWeight[c] > ORC_MINMIX) {
                  int g = m->compGauss[c];
                  if (!doneVa[g]) {
                     float occim = acc->vaOcc[g], x, muDiffk;
                     int mixFloored = 0;
                     if (occim > 0.0) {
                        int shared = ((cfg->uFlags & ORC_UPMEANS) == 0 || doneMu[g] || acc->muOcc[g] <= 0.0);
                        for (k = 0; k < D; k++) {
                           muDiffk = shared ? 0.0 : acc->mu[(size_t)g * D + k] / acc->muOcc[g];
                           x = acc->va[(size_t)g * D + k] / occim - muDiffk * muDiffk;
                           if (x < cfg->minVar) { x = cfg->minVar; st->nFloorVar++; mixFloored = 1; }
                           var[(size_t)g * D + k] = x;
                        }
                     }
                     if (mixFloored) st->nFloorVarMix++;
                     doneVa[g] = 1;
                  }
               }
         }
      /* UpdateMeans HERest.c:974-1012 */
      if (cfg->uFlags & ORC_UPMEANS)
         for (j = 2; j < N; j++) {
            int s = m->hmmState[m->hmmStateOff[h] + (j - 2)];
            for (c = m->stateCompOff[s]; c < m->stateCompOff[s + 1]; c++)
               if (compWeight[c] > ORC_MINMIX) {
                  int g = m->compGauss[c];
                  if (!doneMu[g]) {
                     float occim = acc->muOcc[g];
                     if (occim > 0.0)
                        for (k = 0; k < D; k++)
                           mean[(size_t)g * D + k] += acc->mu[(size_t)g * D + k] / occim;
                     doneMu[g] = 1;
                  }
               }
         }
      /* FixGConsts HModel.c:5688-5714 */
      if (cfg->uFlags & (ORC_UPMEANS | ORC_UPVARS))
         for (j = 2; j < N; j++) {
            int s = m->hmmState[m->hmmStateOff[h] + (j - 2)];
            for (c = m->stateCompOff[s]; c < m->stateCompOff[s + 1]; c++)
               if (compWeight[c] > ORC_MINMIX) {
                  int g = m->compGauss[c];
                  orc_fix_diag_gconst(D, var + (size_t)g * D, gconst + g);
               }
         }
   }
   free(doneT); free(doneS); free(doneMu); free(doneVa); free(occOff);
}
