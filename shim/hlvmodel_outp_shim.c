/* hlvmodel_outp_shim.c -- HDecode's block scorer (HTKLVRec) on top of the MI355X library.
 *
 * HDecode's token pass asks for output probabilities one STATE at a time, a block of frames ahead: on a cache miss cOutP
 * (HLVRec-outP.c:196-262) calls
 *      OutPBlock      (si, obsBlock, n, sIdx, acScale, outP)          HLVModel.c:270, the compact-model path
 *      OutPBlock_HMod (si, obsBlock, n, sIdx, acScale, outP, id)      HLVRec-outP.c:329, USEHMODEL = T
 * and keeps the n scores in its own per-state cache.  This file defines both with the reference's prototypes (HLVModel.h is on the
 * include path at build time only) and serves them from the device: the first call for a block of observations scores ALL tied states
 * of the set for those n frames in one htkamd_outp_block launch (SOutP's arithmetic, the one OutP / POutP_HModel use: HModel.c:5503),
 * keeps the [n x S] table on the host, and every further call for the same block -- the other states the token pass reaches -- is a
 * row copy.  A block is recognised by the content of its first observation.  Link it instead of the two reference definitions
 * (recipe: oracle/Makefile, target _ref/ref_outpblock, which also checks every score against the reference's OutP).
 * Restrictions (HError 7399): one stream, diagonal covariances, PLAINHS / SHAREDHS, no input transform.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "HShell.h"
#include "HMem.h"
#include "HMath.h"
#include "HWave.h"
#include "HLabel.h"
#include "HAudio.h"
#include "HParm.h"
#include "HDict.h"
#include "HModel.h"
#include "HUtil.h"
#include "HLVModel.h"

#include "htk_amd.h"

static struct {
   HMMSet *hset;
   htkamd_model *model;
   int D, S;
   StateInfo **state;            /* [S] tied states in scan order */
   void *dAllStates;             /* 0..S-1, on the device */
   int maxIdx; int *rowOfIdx;    /* StateInfo.sIdx (HDecode's block numbering) -> row */
   float *first;                 /* content of the cached block's first observation */
   int n;                        /* frames in the cached block */
   float *table;                 /* [n][S] */
   float *hX; void *dX, *dOut; int capN;
   int mode;                     /* HTKAMD_SCORE_SOUTP, + HTKAMD_SCORE_DIAGC while the set still holds variances */
} Z;

static void amd_check(int rc, const char *what) { if (rc != HTKAMD_OK) HError(7399, "%s: %s", what, htkamd_last_error()); }

static void pack(HMMSet *hset)
{
   HMMScanState hss;
   int nS = 0, nG = 0, capS = 0, capG = 0, s, g, k, c, C, i, j;
   MixPDF **mix = NULL;
   float *weight, *logwt, *mean, *var, *ivar, *gconst, *transP;
   int inv = 0;
   int *stateCompOff, *compGauss, transN[1] = {3}, transOff[2] = {0, 9}, hmmTrans[1] = {0}, hmmStateOff[2] = {0, 1}, hmmState[1] = {0};
   htkamd_model_desc d;

   if (hset->swidth[0] != 1) HError(7399, "OutPBlock: %d data streams (the MI355X scorer takes one)", hset->swidth[0]);
   if (hset->hsKind != PLAINHS && hset->hsKind != SHAREDHS) HError(7399, "OutPBlock: tied-mixture and discrete systems are not supported");
   if (hset->xf != NULL) HError(7399, "OutPBlock: input transforms are not supported");
   Z.hset = hset; Z.D = hset->vecSize; Z.maxIdx = -1;
   NewHMMScan(hset, &hss);
   while (GoNextState(&hss, FALSE)) {
      if (nS + 1 > capS) { capS = capS * 2 + 256; Z.state = (StateInfo **)realloc(Z.state, sizeof(StateInfo *) * (size_t)capS); }
      Z.state[nS++] = hss.si;
      if (hss.si->sIdx > Z.maxIdx) Z.maxIdx = hss.si->sIdx;
   }
   EndHMMScan(&hss);
   Z.S = nS;
   Z.rowOfIdx = (int *)malloc(sizeof(int) * (size_t)(Z.maxIdx + 2));
   for (i = 0; i <= Z.maxIdx; i++) Z.rowOfIdx[i] = -1;
   for (s = 0; s < nS; s++) if (Z.state[s]->sIdx >= 0) Z.rowOfIdx[Z.state[s]->sIdx] = s;
   stateCompOff = (int *)calloc((size_t)nS + 1, sizeof(int));
   for (s = 0; s < nS; s++) stateCompOff[s + 1] = stateCompOff[s] + Z.state[s]->pdf[1].nMix;
   C = stateCompOff[nS];
   compGauss = (int *)calloc((size_t)C, sizeof(int)); weight = (float *)calloc((size_t)C, sizeof(float)); logwt = (float *)calloc((size_t)C, sizeof(float));
   for (s = 0; s < nS; s++)
      for (k = 1; k <= Z.state[s]->pdf[1].nMix; k++) {
         MixtureElem *me = Z.state[s]->pdf[1].spdf.cpdf + k;
         c = stateCompOff[s] + k - 1;
         if (me->mpdf->ckind != DIAGC && me->mpdf->ckind != INVDIAGC) HError(7399, "OutPBlock: only diagonal covariances are supported");
         for (g = 0; g < nG; g++) if (mix[g] == me->mpdf) break;          /* shared pdfs (~m) are rare: a linear search will do */
         if (g == nG) { if (nG + 1 > capG) { capG = capG * 2 + 1024; mix = (MixPDF **)realloc(mix, sizeof(MixPDF *) * (size_t)capG); } mix[nG++] = me->mpdf; }
         compGauss[c] = g;
         if (hset->logWt) { logwt[c] = me->weight; weight[c] = (me->weight <= LMINMIX) ? 0.0f : (float)exp((double)me->weight); }
         else { weight[c] = me->weight; logwt[c] = (me->weight < MINMIX) ? (float)LZERO : (float)log((double)me->weight); }
      }
   mean = (float *)calloc((size_t)nG * Z.D, sizeof(float)); var = (float *)calloc((size_t)nG * Z.D, sizeof(float)); gconst = (float *)calloc((size_t)nG, sizeof(float));
   ivar = (float *)calloc((size_t)nG * Z.D, sizeof(float));
   for (g = 0; g < nG; g++) {
      if (mix[g]->ckind == INVDIAGC) inv = 1;
      for (k = 1; k <= Z.D; k++) {
         const float v = mix[g]->cov.var[k];
         mean[(size_t)g * Z.D + k - 1] = mix[g]->mean[k];
         if (mix[g]->ckind == INVDIAGC) { ivar[(size_t)g * Z.D + k - 1] = v; var[(size_t)g * Z.D + k - 1] = 1 / v; }
         else { float c2 = v; if (c2 > 1E+30) c2 = 1E+30; if (c2 < 1E-30) c2 = 1E-30; var[(size_t)g * Z.D + k - 1] = v; ivar[(size_t)g * Z.D + k - 1] = 1 / c2; }
      }
      gconst[g] = mix[g]->gConst;
   }
   /* the scorer needs no topology: one dummy three-state model over the first tied state satisfies the description */
   transP = (float *)calloc(9, sizeof(float));
   for (i = 0; i < 9; i++) transP[i] = (float)LZERO;
   transP[1] = 0.0f; transP[4] = (float)log(0.5); transP[5] = (float)log(0.5);
   memset(&d, 0, sizeof(d));
   d.vecSize = Z.D; d.numStates = nS; d.numComp = C; d.numGauss = nG; d.numTrans = 1; d.numPhys = 1;
   d.stateCompOff = stateCompOff; d.compWeight = weight; d.compGauss = compGauss; d.mean = mean; d.var = var; d.gconst = gconst;
   d.transN = transN; d.transOff = transOff; d.transP = transP; d.hmmTrans = hmmTrans; d.hmmStateOff = hmmStateOff; d.hmmState = hmmState;
   (void)j;
   amd_check(htkamd_model_create(&d, &Z.model), "htkamd_model_create");
   if (inv) amd_check(htkamd_model_set_prepared(Z.model, ivar, gconst, logwt), "htkamd_model_set_prepared");   /* the set went through ConvDiagC: its own 1/var */
   Z.mode = HTKAMD_SCORE_SOUTP | (inv ? 0 : HTKAMD_SCORE_DIAGC);
   {
      int *all = (int *)malloc(sizeof(int) * (size_t)nS);
      for (s = 0; s < nS; s++) all[s] = s;
      amd_check(htkamd_dev_malloc(&Z.dAllStates, sizeof(int) * (size_t)nS), "htkamd_dev_malloc");
      amd_check(htkamd_memcpy_h2d(Z.dAllStates, all, sizeof(int) * (size_t)nS, NULL), "htkamd_memcpy_h2d");
      free(all);
   }
   Z.first = (float *)calloc((size_t)Z.D, sizeof(float));
   free(stateCompOff); free(compGauss); free(weight); free(logwt); free(mean); free(var); free(ivar); free(gconst); free(transP); free(mix);
}

static void serve(StateInfo_lv *si, Observation **obsBlock, int n, int sIdx, float acScale, LogFloat *outP)
{
   int i, row;
   if (htkamd_device_count() <= 0) HError(7399, "OutPBlock: %s", "no HIP device: the MI355X scorer has no CPU path (HTKAMD_ENODEV)");
   if (Z.hset != si->hset) { if (Z.hset != NULL) HError(7399, "OutPBlock: one HMM set at a time"); pack(si->hset); }
   row = si->useHModel ? -1 : ((sIdx >= 0 && sIdx <= Z.maxIdx) ? Z.rowOfIdx[sIdx] : -1);
   if (si->useHModel) { for (i = 0; i < Z.S; i++) if (Z.state[i] == si->si[sIdx]) { row = i; break; } }
   if (row < 0) HError(7399, "OutPBlock: state index %d is not a state of the set", sIdx);
   if (n != Z.n || memcmp(Z.first, &obsBlock[0]->fv[1][1], sizeof(float) * (size_t)Z.D) != 0) {
      /* a new block: every tied state for its n frames, once */
      if (n > Z.capN) {
         Z.capN = n;
         Z.hX = (float *)realloc(Z.hX, sizeof(float) * (size_t)n * Z.D); Z.table = (float *)realloc(Z.table, sizeof(float) * (size_t)n * Z.S);
         if (Z.dX) { htkamd_dev_free(Z.dX); htkamd_dev_free(Z.dOut); }
         amd_check(htkamd_dev_malloc(&Z.dX, sizeof(float) * (size_t)n * Z.D), "htkamd_dev_malloc");
         amd_check(htkamd_dev_malloc(&Z.dOut, sizeof(float) * (size_t)n * Z.S), "htkamd_dev_malloc");
      }
      for (i = 0; i < n; i++) memcpy(Z.hX + (size_t)i * Z.D, &obsBlock[i]->fv[1][1], sizeof(float) * (size_t)Z.D);
      amd_check(htkamd_memcpy_h2d(Z.dX, Z.hX, sizeof(float) * (size_t)n * Z.D, NULL), "htkamd_memcpy_h2d");
      amd_check(htkamd_outp_block_mode(Z.model, (const float *)Z.dX, n, (const int *)Z.dAllStates, Z.S, (float *)Z.dOut, n, Z.mode, NULL), "htkamd_outp_block_mode");
      amd_check(htkamd_memcpy_d2h(Z.table, Z.dOut, sizeof(float) * (size_t)n * Z.S, NULL), "htkamd_memcpy_d2h");
      memcpy(Z.first, &obsBlock[0]->fv[1][1], sizeof(float) * (size_t)Z.D);
      Z.n = n;
   }
   for (i = 0; i < n; i++) outP[i] = Z.table[(size_t)row * n + i];                 /* dOut[k*ldo + t], ldo = n */
   if (acScale != 1.0)
      for (i = 0; i < n; i++) outP[i] *= acScale;
}

void OutPBlock(StateInfo_lv *si, Observation **obsBlock, int n, int sIdx, float acScale, LogFloat *outP)
{
   serve(si, obsBlock, n, sIdx, acScale, outP);
}

void OutPBlock_HMod(StateInfo_lv *si, Observation **obsBlock, int n, int sIdx, float acScale, LogFloat *outP, int id)
{
   (void)id;
   serve(si, obsBlock, n, sIdx, acScale, outP);
}
