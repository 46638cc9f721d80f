/* htklib_hfb_shim.c -- HTKLib's forward-backward module (HFB.h) on top of the MI355X library.
 *
 * What this is.  The drop-in boundary of the HERest hot path is HTKLib's link-time C ABI.  This file defines EVERY
 * function HFB.o exports -- InitFB, SetTraceFB, InitialiseForBack, UseAlignHMMSet, InitUttInfo, GetInputObs, LoadLabs, LoadData,
 * InitUttObservations, FBFile, PrLog, SetMinDurs, FindStateOrder (HFB.h:111-150, HFB.c) -- with the reference's prototypes and
 * struct layouts (the reference's own headers are on the include path at build time; nothing of them is kept in this repository),
 * and forwards the numerical work to include/htk_amd.h.  Linking the reference's unchanged HERest.o against
 *       this object + HTKLib (without HFB.o) + libhtk_amd.so      (recipe: oracle/Makefile, target _ref/HERest_amd)
 * gives an HERest whose E-step runs on the GPU and whose own UpdateModels / DumpAccs / StatReport read the statistics from the
 * accumulator hooks as always.
 *
 * How the statistics get back.  They stay on the device over all files of a run (one flat fp64 vector, htkamd_accs) and are
 * added into the reference's accumulators -- MuAcc / VaAcc on the mean and variance hooks, WtAcc on StreamElem.hook, TrAcc on the
 * transP hook, the example counter in HMMDef.hook (HTrain.h:211-232, HFB.c:1768-1772) -- the first time anybody walks the model set
 * afterwards: every consumer (MLUpdateModels HERest.c:1262, DumpAccs HTrain.c:1453, StatReport HERest.c:708) starts with
 * NewHMMScan, so the link line wraps that one symbol (-Wl,--wrap=NewHMMScan) and __wrap_NewHMMScan below flushes first.
 *
 * Restrictions (HError 7399, as the library's own): diagonal covariances, PLAINHS/SHAREDHS (one stream or several), no input transform, no
 * two-model re-estimation, no single-pass retraining (two data files).  FBFile is called per file by the front-end and is served
 * as a batch of one; the batched entry points of htk_amd.h are the fast path (tools/herest.c).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "HShell.h"
#include "HMem.h"
#include "HMath.h"
#include "HSigP.h"
#include "HAudio.h"
#include "HWave.h"
#include "HVQ.h"
#include "HParm.h"
#include "HLabel.h"
#include "HModel.h"
#include "HTrain.h"
#include "HUtil.h"
#include "HAdapt.h"
#include "HFB.h"

#include "htk_amd.h"

#define SHIM_T_TOP 0001

static ConfParam *cParm[MAXGLOBS];
static int nParm = 0;
static int trace = 0;
static struct { LogDouble pruneInit, pruneInc, pruneLim; float minFrwdP; } prune = { NOPRUNE, 0.0, NOPRUNE, 10.0 };

/* ---- device side of one HMMSet ---- */
typedef struct {
   HMMSet *hset;
   int packed;                         /* tables below are valid */
   int D, S, C, G, nT, H;
   int NSt, *dimStream, *gaussStream;  /* data streams (hset->swidth[0]); dimension -> stream of the undivided row; stream of Gaussian g (1-based g) */
   HLink *hmmOf;                       /* [H] */
   StreamElem **steOf;                 /* [S] FIRST stream element of tied state s (its streams follow it) */
   MixPDF **mixOf;                     /* [G] */
   HLink *transOwner;                  /* [nT] a model that owns matrix t */
   int *stateCompOff, *compGauss, *transN, *transOff, *hmmTrans, *hmmStateOff, *hmmState;
   /* pointer -> index tables (open addressing) */
   void **hkey; int *hval; size_t hcap, hn;
   htkamd_model *model;
   htkamd_accs *accs;
   htkamd_fb *fb;
   void *dX; size_t dXcap;
   float *hX; size_t hXcap;            /* page-locked staging of an utterance's observations */
   int *labs; size_t labsCap;
   int exactLadd;                      /* HTKAMD_SHIM_EXACT=1: the table-driven log-add in the recursions (alpha / beta bit-compatible); default: fp32 transcendentals (tolerance class) */
   int dirty;                          /* statistics on the device not yet added to the hooks */
   UPDSet uFlags;
} ShimSet;

static ShimSet g_set;                   /* HFB serves one set at a time (file-scope state there too) */

static void amd_check(int rc, const char *what)
{
   if (rc != HTKAMD_OK) HError(7399, "%s: %s", what, htkamd_last_error());
}

static void map_put(ShimSet *z, void *key, int val)
{
   size_t h;
   if (2 * (z->hn + 1) > z->hcap) {                    /* grow: keep the table at most half full */
      void **ok = z->hkey; int *ov = z->hval; const size_t oc = z->hcap;
      size_t i;
      z->hcap = oc ? 2 * oc : 1024; z->hn = 0;
      z->hkey = (void **)calloc(z->hcap, sizeof(void *)); z->hval = (int *)calloc(z->hcap, sizeof(int));
      for (i = 0; i < oc; i++) if (ok[i] != NULL) map_put(z, ok[i], ov[i]);
      free(ok); free(ov);
   }
   h = ((size_t)key >> 4) * 0x9E3779B97F4A7C15ull % z->hcap;
   while (z->hkey[h] != NULL && z->hkey[h] != key) h = (h + 1) % z->hcap;
   if (z->hkey[h] == NULL) z->hn++;
   z->hkey[h] = key; z->hval[h] = val;
}

static int map_get(const ShimSet *z, const void *key)
{
   size_t h;
   if (z->hcap == 0) return -1;
   h = ((size_t)key >> 4) * 0x9E3779B97F4A7C15ull % z->hcap;
   while (z->hkey[h] != NULL) {
      if (z->hkey[h] == key) return z->hval[h];
      h = (h + 1) % z->hcap;
   }
   return -1;
}

/* ------------------------------------------------------------------------------------------------ module initialisation */

static char *shim_version = "!HVER!HFB(htk_amd shim):   3.4.1 [MI355X]";
static char *shim_vc_id = "htk_amd";

void InitFB(void)
{
   int m, i;
   double d;

   Register(shim_version, shim_vc_id);
   for (m = 0; m < 2; ++m) {
      nParm = GetConfig(m == 0 ? "HFWDBKWD" : "HFB", TRUE, cParm, MAXGLOBS);
      if (nParm > 0) {
         if (GetConfInt(cParm, nParm, "TRACE", &i)) trace = i;
         if (GetConfFlt(cParm, nParm, "PRUNEINIT", &d)) prune.pruneInit = d;
         if (GetConfFlt(cParm, nParm, "PRUNEINC", &d)) prune.pruneInc = d;
         if (GetConfFlt(cParm, nParm, "PRUNELIM", &d)) prune.pruneLim = d;
         if (GetConfFlt(cParm, nParm, "MINFORPROB", &d)) prune.minFrwdP = d;
      }
   }
}

void SetTraceFB(void) { trace |= SHIM_T_TOP; }

void PrLog(LogDouble x)
{
   if (x < LSMALL) printf("       LZERO");
   else printf("%12.5f", x);
}

/* SetMinDurs / FindStateOrder are exported by HFB.o; nothing outside it calls them, the library derives minimum durations itself
   (htk_amd/host/prep.c).  Kept so that the symbol set of the module is complete. */
void SetMinDurs(HMMSet *hset) { (void)hset; }
void FindStateOrder(HLink hmm, IntVec so, int s, int *d) { (void)hmm; (void)so; (void)s; (void)d; }

/* weight / transition accumulators and the example counter, as HFB re-attaches them on every InitialiseForBack (HFB.c:187-210) */
static void attach_counters(HMMSet *hset, MemHeap *x)
{
   HMMScanState hss;

   NewHMMScan(hset, &hss);
   do {
      HLink hmm = hss.hmm;
      hmm->hook = (void *)0;
      if (!IsSeenV(hmm->transP)) {
         TrAcc *ta = (TrAcc *)New(x, sizeof(TrAcc));
         ta->tran = CreateMatrix(x, hmm->numStates, hmm->numStates); ZeroMatrix(ta->tran);
         ta->occ = CreateVector(x, hmm->numStates); ZeroVector(ta->occ);
         SetHook(hmm->transP, ta);
         TouchV(hmm->transP);
      }
      while (GoNextState(&hss, TRUE))
         while (GoNextStream(&hss, TRUE)) {
            WtAcc *wa = (WtAcc *)New(x, sizeof(WtAcc));
            wa->c = CreateVector(x, hss.M); ZeroVector(wa->c);
            wa->occ = 0.0; wa->time = -1; wa->prob = NULL;
            hss.ste->hook = wa;
         }
   } while (GoNextHMM(&hss));
   EndHMMScan(&hss);
}

void InitialiseForBack(FBInfo *fbInfo, MemHeap *x, HMMSet *hset, UPDSet uset,
                       LogDouble pruneInit, LogDouble pruneInc, LogDouble pruneLim, float minFrwdP)
{
   int s;

   if (htkamd_device_count() <= 0)
      HError(7399, "InitialiseForBack: %s", "no HIP device: the MI355X forward-backward module has no CPU path (HTKAMD_ENODEV)");
   fbInfo->uFlags = uset;
   fbInfo->up_hset = fbInfo->al_hset = hset;
   fbInfo->twoModels = FALSE;
   fbInfo->hsKind = hset->hsKind;
   attach_counters(hset, x);
   fbInfo->maxM = MaxMixInSet(hset);
   fbInfo->skipstart = fbInfo->skipend = -1;
   for (s = 1; s <= hset->swidth[0]; s++) fbInfo->maxMixInS[s] = MaxMixInSetS(hset, s);
   fbInfo->ab = (AlphaBeta *)New(x, sizeof(AlphaBeta));
   memset(fbInfo->ab, 0, sizeof(AlphaBeta));
   CreateHeap(&fbInfo->ab->abMem, "AlphaBetaFB", MSTAK, 1, 1.0, 1000, 10000);
   if (pruneInit < NOPRUNE) { prune.pruneInit = pruneInit; prune.pruneInc = pruneInc; prune.pruneLim = pruneLim; }
   if (minFrwdP < NOPRUNE) prune.minFrwdP = minFrwdP;
   if (prune.pruneInit < NOPRUNE) {
      if (prune.pruneInc != 0.0) printf("Pruning-On[%.1f %.1f %.1f]\n", prune.pruneInit, prune.pruneInc, prune.pruneLim);
      else printf("Pruning-On[%.1f]\n", prune.pruneInit);
   } else printf("Pruning-Off\n");
   if (hset->hsKind != PLAINHS && hset->hsKind != SHAREDHS) HError(7399, "InitialiseForBack: tied-mixture and discrete systems are not supported by this module (tools/herest serves <TMIX> sets)");
   memset(&g_set, 0, sizeof(g_set));
   g_set.hset = hset; g_set.uFlags = uset;
}

void UseAlignHMMSet(FBInfo *fbInfo, MemHeap *x, HMMSet *al_hset)
{
   (void)fbInfo; (void)x; (void)al_hset;
   HError(7399, "UseAlignHMMSet: two-model re-estimation is not supported by the MI355X forward-backward module");
}

void InitUttInfo(UttInfo *utt, Boolean twoFiles)
{
   CreateHeap(&utt->transStack, "transStore", MSTAK, 1, 0.5, 1000, 10000);
   CreateHeap(&utt->dataStack, "dataStore", MSTAK, 1, 0.5, 1000, 10000);
   if (twoFiles) CreateHeap(&utt->dataStack2, "dataStore2", MSTAK, 1, 0.5, 1000, 10000);
   utt->pbuf = NULL; utt->pbuf2 = NULL; utt->tr = NULL;
}

void GetInputObs(UttInfo *utt, int t, HSetKind hsKind)
{
   (void)hsKind;
   ReadAsTable(utt->pbuf, t - 1, &utt->ot);
}

void LoadLabs(UttInfo *utt, FileFormat lff, char *datafn, char *labDir, char *labExt)
{
   char labfn[MAXFNAMELEN], b1[MAXFNAMELEN], b2[MAXFNAMELEN];

   ResetHeap(&utt->transStack);
   MakeFN(datafn, labDir, labExt, labfn);
   if (trace & SHIM_T_TOP) {
      printf(" Processing Data: %s; Label %s\n", NameOf(datafn, b1), NameOf(labfn, b2));
      fflush(stdout);
   }
   utt->tr = LOpen(&utt->transStack, labfn, lff);
   utt->Q = CountLabs(utt->tr->head);
   if (utt->Q == 0) HError(-7325, "LoadUtterance: No labels in file %s", labfn);
}

void LoadData(HMMSet *hset, UttInfo *utt, FileFormat dff, char *datafn, char *datafn2)
{
   BufferInfo info;

   (void)datafn2;
   if (utt->twoDataFiles) HError(7399, "LoadData: single-pass retraining (two data files) is not supported by the MI355X module");
   if (utt->pbuf != NULL) CloseBuffer(utt->pbuf);
   ResetHeap(&utt->dataStack);
   if ((utt->pbuf = OpenBuffer(&utt->dataStack, datafn, 0, dff, FALSE_dup, FALSE_dup)) == NULL)
      HError(7350, "HFB: Config parameters invalid");
   GetBufferInfo(utt->pbuf, &info);
   if (info.tgtVecSize != hset->vecSize)
      HError(7350, "CheckData: Vector size in %s[%d] is incompatible with hset [%d]", datafn, info.tgtVecSize, hset->vecSize);
   if (info.tgtPK != hset->pkind)
      HError(7350, "CheckData: Parameterisation in %s is incompatible with hset", datafn);
   utt->T = ObsInBuffer(utt->pbuf);
}

void InitUttObservations(UttInfo *utt, HMMSet *al_hset, char *datafn, int *maxMixInS)
{
   BufferInfo info;
   Boolean eSep;

   (void)datafn; (void)maxMixInS;
   GetBufferInfo(utt->pbuf, &info);
   SetStreamWidths(info.tgtPK, info.tgtVecSize, al_hset->swidth, &eSep);
   utt->ot = MakeObservation(&gstack, al_hset->swidth, info.tgtPK, FALSE, eSep);
}

/* ------------------------------------------------------------------------------------------------ HMMSet -> packed model */

static void pack_set(ShimSet *z)
{
   HMMSet *hset = z->hset;
   HMMScanState hss;
   int h, s, c, g, t, i, j, k;
   htkamd_model_desc d;
   float *weight, *mean, *var, *gconst, *transP, *ivar, *logwt;
   int nTp = 0, nHs = 0;

   if (hset->xf != NULL) HError(7399, "FBFile: input transforms are not supported by the MI355X module");
   /* Own numbering of the shared structures (a pointer -> index table): physical models in scan order, tied states, Gaussians and
      transition matrices in the order they are first met.  HModel's SetIndexes (HModel.c:3942) is NOT used for this: it borrows the
      hook of every transition matrix and leaves it NULL -- the hook that carries the TrAcc the front-end's UpdateTrans reads. */
   z->H = hset->numPhyHMM;
   z->hmmOf = (HLink *)calloc((size_t)z->H, sizeof(HLink));
   {
      int capS = 1024, capG = 4096, capT = 256;
      z->steOf = (StreamElem **)calloc((size_t)capS + 1, sizeof(StreamElem *));
      z->mixOf = (MixPDF **)calloc((size_t)capG + 1, sizeof(MixPDF *));
      z->transOwner = (HLink *)calloc((size_t)capT + 1, sizeof(HLink));
      z->gaussStream = (int *)calloc((size_t)capG + 1, sizeof(int));
      z->S = z->G = z->nT = 0;
      z->NSt = hset->swidth[0];
      z->D = hset->vecSize;
      z->dimStream = (int *)calloc((size_t)z->D, sizeof(int));
      if (z->NSt > 1) {                                   /* where ExtractObservation puts each element of the row (HParm.c:2843) */
         char kind[64], why[160];
         int w[SMAX];
         for (s = 1; s <= z->NSt; s++) w[s - 1] = hset->swidth[s];
         ParmKind2Str(hset->pkind, kind);
         if (htkamd_host_stream_dims(kind, z->D, z->NSt, w, z->dimStream, why, sizeof(why))) HError(7399, "FBFile: stream widths: %s", why);
      }
      h = 0;
      NewHMMScan(hset, &hss);
      do {
         HLink hmm = hss.hmm;
         if (h >= z->H) HError(7399, "FBFile: more physical models scanned than the set declares (%d)", z->H);
         z->hmmOf[h] = hmm; map_put(z, hmm, h);
         if (map_get(z, hmm->transP) < 0) {
            if (z->nT + 1 > capT) { capT *= 2; z->transOwner = (HLink *)realloc(z->transOwner, sizeof(HLink) * ((size_t)capT + 1)); }
            map_put(z, hmm->transP, z->nT); z->transOwner[++z->nT] = hmm;
            nTp += hmm->numStates * hmm->numStates;
         }
         nHs += hmm->numStates - 2;
         for (j = 2; j < hmm->numStates; j++) {
            StateInfo *si = hmm->svec[j].info;
            StreamElem *ste = si->pdf + 1;
            if (map_get(z, si) >= 0) continue;
            if (z->S + 1 > capS) { capS *= 2; z->steOf = (StreamElem **)realloc(z->steOf, sizeof(StreamElem *) * ((size_t)capS + 1)); }
            map_put(z, si, z->S); z->steOf[++z->S] = ste;
            for (s = 0; s < z->NSt; s++)
            for (k = 1; k <= ste[s].nMix; k++) {
               MixPDF *mp = ste[s].spdf.cpdf[k].mpdf;
               if (map_get(z, mp) >= 0) continue;
               if (z->G + 1 > capG) { capG *= 2; z->mixOf = (MixPDF **)realloc(z->mixOf, sizeof(MixPDF *) * ((size_t)capG + 1)); z->gaussStream = (int *)realloc(z->gaussStream, sizeof(int) * ((size_t)capG + 1)); }
               map_put(z, mp, z->G); z->mixOf[++z->G] = mp; z->gaussStream[z->G] = s;
            }
         }
         h++;
      } while (GoNextHMM(&hss));
      EndHMMScan(&hss);
   }
   if (h != z->H) HError(7399, "FBFile: %d physical models scanned, %d expected", h, z->H);
   /* one entry per (state, stream): element e = (s - 1) * NSt + stream */
   z->stateCompOff = (int *)calloc((size_t)z->S * z->NSt + 1, sizeof(int));
   for (s = 1; s <= z->S; s++) for (k = 0; k < z->NSt; k++) { const int e = (s - 1) * z->NSt + k; z->stateCompOff[e + 1] = z->stateCompOff[e] + z->steOf[s][k].nMix; }
   z->C = z->stateCompOff[z->S * z->NSt];
   z->compGauss = (int *)calloc((size_t)z->C, sizeof(int));
   weight = (float *)calloc((size_t)z->C, sizeof(float)); logwt = (float *)calloc((size_t)z->C, sizeof(float));
   for (s = 1; s <= z->S; s++)
      for (i = 0; i < z->NSt; i++)
      for (k = 1; k <= z->steOf[s][i].nMix; k++) {
         MixtureElem *me = z->steOf[s][i].spdf.cpdf + k;
         MixPDF *mp = me->mpdf;
         c = z->stateCompOff[(s - 1) * z->NSt + i] + k - 1;
         if (mp->ckind != DIAGC && mp->ckind != INVDIAGC) HError(7399, "FBFile: only diagonal covariances are supported by the MI355X module");
         z->compGauss[c] = map_get(z, mp);
         /* the front-end has already run ConvLogWt when the first file arrives (HERest.c:640): keep BOTH forms exact */
         if (hset->logWt) { logwt[c] = me->weight; weight[c] = (me->weight <= LMINMIX) ? 0.0f : (float)exp((double)me->weight); }
         else { weight[c] = me->weight; logwt[c] = (me->weight < MINMIX) ? (float)LZERO : (float)log((double)me->weight); }
      }
   mean = (float *)calloc((size_t)z->G * z->D, sizeof(float)); var = (float *)calloc((size_t)z->G * z->D, sizeof(float));
   ivar = (float *)calloc((size_t)z->G * z->D, sizeof(float)); gconst = (float *)calloc((size_t)z->G, sizeof(float));
   for (g = 1; g <= z->G; g++) {
      MixPDF *mp = z->mixOf[g];
      int kk = 0;                                          /* index in the stream's own vector */
      for (k = 1; k <= z->D; k++) {
         const size_t at = (size_t)(g - 1) * z->D + k - 1;
         if (z->NSt > 1 && z->dimStream[k - 1] != z->gaussStream[g]) { mean[at] = 0.0f; var[at] = INFINITY; ivar[at] = 0.0f; continue; }   /* include/htk_amd.h: undivided rows */
         kk++;
         const float v = mp->cov.var[kk];
         mean[at] = mp->mean[kk];
         if (mp->ckind == INVDIAGC) { ivar[at] = v; var[at] = 1 / v; }
         else { float c2 = v; if (c2 > 1E+30) c2 = 1E+30; if (c2 < 1E-30) c2 = 1E-30; var[at] = v; ivar[at] = 1 / c2; }
      }
      gconst[g - 1] = mp->gConst;
   }
   z->transN = (int *)calloc((size_t)z->nT, sizeof(int)); z->transOff = (int *)calloc((size_t)z->nT + 1, sizeof(int));
   transP = (float *)calloc((size_t)nTp, sizeof(float));
   for (t = 1; t <= z->nT; t++) {
      HLink hmm = z->transOwner[t];
      const int N = hmm->numStates;
      z->transN[t - 1] = N; z->transOff[t] = z->transOff[t - 1] + N * N;
      for (i = 1; i <= N; i++)
         for (j = 1; j <= N; j++) transP[z->transOff[t - 1] + (i - 1) * N + (j - 1)] = hmm->transP[i][j];
   }
   z->hmmTrans = (int *)calloc((size_t)z->H, sizeof(int)); z->hmmStateOff = (int *)calloc((size_t)z->H + 1, sizeof(int));
   z->hmmState = (int *)calloc((size_t)nHs, sizeof(int));
   for (h = 0; h < z->H; h++) {
      HLink hmm = z->hmmOf[h];
      z->hmmTrans[h] = map_get(z, hmm->transP);
      z->hmmStateOff[h + 1] = z->hmmStateOff[h] + hmm->numStates - 2;
      for (j = 2; j < hmm->numStates; j++) z->hmmState[z->hmmStateOff[h] + j - 2] = map_get(z, hmm->svec[j].info);
   }
   memset(&d, 0, sizeof(d));
   d.vecSize = z->D; d.numStates = z->S; d.numComp = z->C; d.numGauss = z->G; d.numTrans = z->nT; d.numPhys = z->H;
   d.stateCompOff = z->stateCompOff; d.compWeight = weight; d.compGauss = z->compGauss; d.mean = mean; d.var = var; d.gconst = gconst;
   d.transN = z->transN; d.transOff = z->transOff; d.transP = transP; d.hmmTrans = z->hmmTrans; d.hmmStateOff = z->hmmStateOff; d.hmmState = z->hmmState;
   d.numStreams = z->NSt; d.dimStream = z->NSt > 1 ? z->dimStream : NULL;
   amd_check(htkamd_model_create(&d, &z->model), "htkamd_model_create");
   /* the tables the kernels read are the front-end's own numbers, bit for bit: 1/variance from ConvDiagC, log weights from ConvLogWt */
   amd_check(htkamd_model_set_prepared(z->model, ivar, gconst, logwt), "htkamd_model_set_prepared");
   /* under the reference's own HERest.o the SEMANTICS are the reference's, defects included: Setotprob's second visit of a tied state as HFB.c:1059 has it */
   amd_check(htkamd_model_set_compat(z->model, HTKAMD_COMPAT_STREAM_REVISIT), "htkamd_model_set_compat");
   /* The recursions: state scores are ALWAYS the reference's (SCORE_EXACT); the log-adds of alpha / beta come from the fp32 transcendental unit by
      default (tolerance class: every re-estimated parameter within 1e-4; 0.5 ms per FBFile call) and from the table-driven exact log-add under
      HTKAMD_SHIM_EXACT=1 in the environment (alpha / beta / pr bit-compatible with the reference's, 3.2 ms per call) -- measured by tools/shim_latency.py */
   { const char *x = getenv("HTKAMD_SHIM_EXACT"); z->exactLadd = (x != NULL && x[0] == '1'); }
   amd_check(htkamd_accs_create(z->model, &z->accs), "htkamd_accs_create");
   amd_check(htkamd_fb_create(z->model, &z->fb), "htkamd_fb_create");
   free(weight); free(logwt); free(mean); free(var); free(ivar); free(gconst); free(transP);
   z->packed = 1;
}

/* ------------------------------------------------------------------------------------------------ statistics -> hooks */

static void flush_to_hooks(ShimSet *z)
{
   htkamd_accs_layout lay;
   double *v;
   int h, s, g, t, i, j, k;

   if (!z->packed || !z->dirty) return;
   z->dirty = 0;
   amd_check(htkamd_accs_get_layout(z->accs, &lay), "htkamd_accs_get_layout");
   v = (double *)malloc(sizeof(double) * lay.total);
   amd_check(htkamd_accs_download(z->accs, v, NULL), "htkamd_accs_download");
   amd_check(htkamd_accs_zero(z->accs, NULL), "htkamd_accs_zero");
   if (getenv("HTKAMD_SHIM_DUMP")) {                    /* debugging aid: the raw vector of this flush */
      FILE *df = fopen(getenv("HTKAMD_SHIM_DUMP"), "ab");
      if (df) { fwrite(v, sizeof(double), lay.total, df); fclose(df); }
   }
   for (h = 0; h < z->H; h++) {
      const long n = (long)z->hmmOf[h]->hook + (long)llround(v[lay.nEgs + h]);
      z->hmmOf[h]->hook = (void *)n;
   }
   for (t = 1; t <= z->nT; t++) {
      HLink hmm = z->transOwner[t];
      TrAcc *ta = (TrAcc *)GetHook(hmm->transP);
      const int N = hmm->numStates;
      int occOff = 0;
      if (ta == NULL) continue;
      for (i = 1; i < t; i++) occOff += z->transN[i - 1];
      for (i = 1; i <= N; i++) {
         for (j = 1; j <= N; j++) ta->tran[i][j] += (float)v[lay.tr + z->transOff[t - 1] + (i - 1) * N + (j - 1)];
         ta->occ[i] += (float)v[lay.trOcc + occOff + i - 1];
      }
   }
   for (s = 1; s <= z->S; s++)
      for (i = 0; i < z->NSt; i++) {
         const int e = (s - 1) * z->NSt + i;
         WtAcc *wa = (WtAcc *)z->steOf[s][i].hook;
         if (wa == NULL) continue;
         for (k = 1; k <= z->steOf[s][i].nMix; k++) wa->c[k] += (float)v[lay.wt + z->stateCompOff[e] + k - 1];
         wa->occ += (float)v[lay.wtOcc + e];
      }
   for (g = 1; g <= z->G; g++) {
      MixPDF *mp = z->mixOf[g];
      MuAcc *ma = (z->uFlags & UPMEANS) ? (MuAcc *)GetHook(mp->mean) : NULL;
      VaAcc *va = (z->uFlags & UPVARS) ? (VaAcc *)GetHook(mp->cov.var) : NULL;
      if (ma != NULL) {
         int kk = 0;
         for (k = 1; k <= z->D; k++) if (z->NSt == 1 || z->dimStream[k - 1] == z->gaussStream[g]) ma->mu[++kk] += (float)v[lay.mu + (size_t)(g - 1) * z->D + k - 1];
         ma->occ += (float)v[lay.muOcc + g - 1];
      }
      if (va != NULL) {
         int kk = 0;
         for (k = 1; k <= z->D; k++) if (z->NSt == 1 || z->dimStream[k - 1] == z->gaussStream[g]) va->cov.var[++kk] += (float)v[lay.va + (size_t)(g - 1) * z->D + k - 1];
         va->occ += (float)v[lay.vaOcc + g - 1];
      }
   }
   free(v);
}

/* every reader of the accumulators starts with an HMM scan: bring them up to date first (link with -Wl,--wrap=NewHMMScan) */
void __real_NewHMMScan(HMMSet *hset, HMMScanState *hss);
void __wrap_NewHMMScan(HMMSet *hset, HMMScanState *hss)
{
   if (g_set.dirty && hset == g_set.hset) flush_to_hooks(&g_set);
   __real_NewHMMScan(hset, hss);
}

/* ------------------------------------------------------------------------------------------------ FBFile */

Boolean FBFile(FBInfo *fbInfo, UttInfo *utt, char *datafn)
{
   ShimSet *z = &g_set;
   htkamd_batch_desc b;
   htkamd_fb_config cfg;
   int frameOff[2], labOff[2], *labs, q, t, k, status = 0;
   double pr = LZERO;
   LLink lab;
   const char *fn = datafn ? datafn : "(buffer)";

   if (fbInfo->al_hset != z->hset) HError(7399, "FBFile: model set differs from the one given to InitialiseForBack");
   if (!z->packed) pack_set(z);
   /* CreateInsts (HFB.c:508): label sequence -> physical models */
   if ((size_t)utt->Q + 1 > z->labsCap) { free(z->labs); z->labsCap = 2 * (size_t)utt->Q + 64; z->labs = (int *)malloc(sizeof(int) * z->labsCap); }
   labs = z->labs;
   for (q = 1; q <= utt->Q; q++) {
      MLink ml;
      lab = GetLabN(utt->tr->head, q);
      if ((ml = FindMacroName(z->hset, 'l', lab->labid)) == NULL) HError(7321, "CreateInsts: Unknown label %s", lab->labid->name);
      labs[q - 1] = map_get(z, ml->structure);
      if (labs[q - 1] < 0) HError(7321, "CreateInsts: label %s has no physical model in the packed set", lab->labid->name);
   }
   /* observations o_1..o_T as one row-major matrix */
   if ((size_t)utt->T * z->D > z->hXcap) {               /* page-locked: the copy below is a DMA that nobody waits for */
      if (z->hX) amd_check(htkamd_host_free(z->hX), "htkamd_host_free");
      z->hXcap = (size_t)utt->T * z->D * 2;
      amd_check(htkamd_host_malloc((void **)&z->hX, sizeof(float) * z->hXcap), "htkamd_host_malloc");
   }
   for (t = 0; t < utt->T; t++) {
      ReadAsTable(utt->pbuf, t, &utt->ot);
      if (z->NSt == 1) for (k = 1; k <= z->D; k++) z->hX[(size_t)t * z->D + k - 1] = utt->ot.fv[1][k];
      else {                                              /* the undivided row back from the stream vectors */
         int at[SMAX] = {0};
         for (k = 0; k < z->D; k++) { const int st = z->dimStream[k]; z->hX[(size_t)t * z->D + k] = utt->ot.fv[st + 1][++at[st]]; }
      }
   }
   if (sizeof(float) * (size_t)utt->T * z->D > z->dXcap) {
      if (z->dX) amd_check(htkamd_dev_free(z->dX), "htkamd_dev_free");
      z->dXcap = sizeof(float) * (size_t)utt->T * z->D * 2;
      amd_check(htkamd_dev_malloc(&z->dX, z->dXcap), "htkamd_dev_malloc");
   }
   amd_check(htkamd_memcpy_h2d_async(z->dX, z->hX, sizeof(float) * (size_t)utt->T * z->D, NULL), "htkamd_memcpy_h2d_async");      /* (htkamd_fb_results below waits for the pass behind it) */
   frameOff[0] = 0; frameOff[1] = utt->T; labOff[0] = 0; labOff[1] = utt->Q;
   b.nUtt = 1; b.dX = (const float *)z->dX; b.frameOff = frameOff; b.labOff = labOff; b.labs = labs;
   cfg.pruneInit = prune.pruneInit; cfg.pruneInc = prune.pruneInc; cfg.pruneLim = prune.pruneLim;
   /* One utterance per call: the call's time is the LATENCY of one utterance's recursions, T dependent steps each.  The state scores are exact
      (SCORE_EXACT: IDOutP's arithmetic) in either mode; the recursions' log-adds: see pack_set (default fp32 transcendentals, HTKAMD_SHIM_EXACT=1 the table) */
   cfg.minFrwdP = prune.minFrwdP; cfg.uFlags = 0; cfg.scoreMode = z->exactLadd ? HTKAMD_SCORE_EXACT : (HTKAMD_SCORE_EXACT | HTKAMD_SCORE_FASTLADD);
   if (fbInfo->uFlags & UPMEANS) cfg.uFlags |= HTKAMD_UPMEANS;
   if (fbInfo->uFlags & UPVARS) cfg.uFlags |= HTKAMD_UPVARS;
   if (fbInfo->uFlags & UPTRANS) cfg.uFlags |= HTKAMD_UPTRANS;
   if (fbInfo->uFlags & UPMIXES) cfg.uFlags |= HTKAMD_UPMIXES;
   amd_check(htkamd_fb_prepare(z->fb, &b, NULL), "htkamd_fb_prepare");
   amd_check(htkamd_fb_execute(z->fb, &cfg, z->accs, NULL), "htkamd_fb_execute");
   amd_check(htkamd_fb_results(z->fb, &pr, &status, NULL), "htkamd_fb_results");
   z->dirty = 1;
   if (status == HTKAMD_UTT_ETEE) HError(7332, "CreateInsts: Cannot have successive Tee models or Tee models at start or end of transcription");
   if (status == HTKAMD_UTT_EALPHA) HError(7390, "StepAlpha: Alpha prune failed");
   if (status != HTKAMD_UTT_OK) {
      if (trace & SHIM_T_TOP) printf(" No path found in beta pass\n");
      HError(-7324, "StepBack: File %s - bad data or over pruning\n", fn);
      return FALSE;
   }
   utt->pr = pr;
   if (trace & SHIM_T_TOP) printf(" Utterance prob per frame = %e\n", utt->pr / utt->T);
   return TRUE;
}
