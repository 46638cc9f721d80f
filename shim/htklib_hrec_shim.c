/* htklib_hrec_shim.c -- HTKLib's recogniser entry points (HRec.h:150-190) on top of the MI355X library, for the reference's HVite.
 *
 * HVite drives the recogniser as  StartRecognition -> ProcessObservation (once per frame) -> CompleteRecognition -> Lattice,
 * and turns the lattice into labels with TranscriptionFromLattice / FormatTranscription.  Linking the reference's UNCHANGED HVite.o
 * with  -Wl,--wrap=StartRecognition -Wl,--wrap=ProcessObservation -Wl,--wrap=CompleteRecognition -Wl,--wrap=InitPSetInfo
 * -Wl,--wrap=InitVRecInfo  sends those five calls here and leaves everything else of HTKLib -- including HRec.o's own
 * TranscriptionFromLattice / FormatTranscription, HNet's network expansion, HParm's buffers -- as it is:
 *   InitPSetInfo        remembers the HMMSet (the rest is the reference's)
 *   InitVRecInfo        remembers nToks / models / states (the rest is the reference's); model/state-level alignment is refused here
 *                       (HError 7399) -- tools/hvite covers it; nToks > 1 (HVite -n) runs the token-set kernel (htkamd_decoder_run_lattice)
 *   StartRecognition    Network (NetNode graph, HNet.h) -> htkamd_net_desc, model set -> htkamd_model, decoder (cached per network)
 *   ProcessObservation  appends the frame to a host table
 *   CompleteRecognition uploads the table, runs the batch-of-one decoder (htkamd_decoder_run_out) and builds the Lattice exactly as
 *                       CreateLattice / LatFromPaths (HRec.c:1679,1512) do for the 1-best path: node 0 = start, node 1 = the end of
 *                       the utterance, nodes 2.. = word ends from the last word back; arcs carry aclike / lmlike / prlike / score.
 * Token likelihoods, times and scores are the reference's bit for bit (htkamd decoder contract), so the label files are HVite's.
 *
 * Restrictions (HError 7399): one stream, diagonal covariances, no input transform, at most 8 tokens per state, word-level output, no tagged
 * null nodes (sub-lattice tags).  Built only where the reference's headers are (oracle/Makefile, target _ref/HVite_amd).
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
#include "HUtil.h"
#include "HTrain.h"
#include "HAdapt.h"
#include "HDict.h"
#include "HNet.h"
#include "HRec.h"

#include "htk_amd.h"

/* ---- pointer -> index table ---- */
typedef struct { void **k; int *v; size_t cap, n; } pmap;
static void pm_put(pmap *m, void *key, int val)
{
   size_t h;
   if (2 * (m->n + 1) > m->cap) {
      void **ok = m->k; int *ov = m->v; const size_t oc = m->cap; size_t i;
      m->cap = oc ? 2 * oc : 1024; m->n = 0;
      m->k = (void **)calloc(m->cap, sizeof(void *)); m->v = (int *)calloc(m->cap, sizeof(int));
      for (i = 0; i < oc; i++) if (ok[i]) pm_put(m, ok[i], ov[i]);
      free(ok); free(ov);
   }
   h = ((size_t)key >> 4) * 0x9E3779B97F4A7C15ull % m->cap;
   while (m->k[h] && m->k[h] != key) h = (h + 1) % m->cap;
   if (!m->k[h]) m->n++;
   m->k[h] = key; m->v[h] = val;
}
static int pm_get(const pmap *m, const void *key)
{
   size_t h;
   if (!m->cap) return -1;
   h = ((size_t)key >> 4) * 0x9E3779B97F4A7C15ull % m->cap;
   while (m->k[h]) { if (m->k[h] == key) return m->v[h]; h = (h + 1) % m->cap; }
   return -1;
}
static void pm_free(pmap *m) { free(m->k); free(m->v); memset(m, 0, sizeof(*m)); }

static void amd_check(int rc, const char *what) { if (rc != HTKAMD_OK) HError(7399, "%s: %s", what, htkamd_last_error()); }

/* ---- state of the shim (HRec keeps file-scope state too: one recogniser at a time, HRec.c:253) ---- */
static struct {
   HMMSet *hset;
   int nToks, models, states;
   htkamd_model *model; pmap hmmIdx;           /* HLink -> physical index */
   int NSt, *dimStream;                        /* data streams; dimension -> stream of the undivided feature row (several streams) */
   Network *net; htkamd_decoder *dec;          /* decoder of the network last started */
   NetNode **wordNode; int nWordNode;          /* WORD node table: htkamd "pronunciation" index -> NetNode */
   float scale, wordpen, pscale;
   float *X; size_t nX, capX; int D;
} S;

static void pack_model(void)
{
   HMMSet *hset = S.hset;
   HMMScanState hss;
   pmap sm = {0}, gm = {0}, tm = {0};
   StreamElem **ste = NULL; MixPDF **mix = NULL; HLink *towner = NULL;
   int nS = 0, nG = 0, nT = 0, capS = 0, capG = 0, capT = 0, H = hset->numPhyHMM, h = 0, s, g, t, i, j, k, c, nTp = 0, nHs = 0;
   HLink *hmmOf = (HLink *)calloc((size_t)H, sizeof(HLink));
   htkamd_model_desc d;
   float *weight, *logwt, *mean, *var, *ivar, *gconst, *transP;
   int *stateCompOff, *compGauss, *transN, *transOff, *hmmTrans, *hmmStateOff, *hmmState, C;
   int *gStr = NULL, ks;
   float *sweight = NULL;
   const int NSt = hset->swidth[0];

   S.NSt = NSt; S.D = hset->vecSize;
   free(S.dimStream); S.dimStream = (int *)calloc((size_t)S.D, sizeof(int));
   if (NSt > 1) {                                          /* where ExtractObservation puts each element of the row (HParm.c:2843) */
      char kind[64], why[160];
      int w[SMAX];
      for (s = 1; s <= NSt; s++) w[s - 1] = hset->swidth[s];
      ParmKind2Str(hset->pkind, kind);
      if (htkamd_host_stream_dims(kind, S.D, NSt, w, S.dimStream, why, sizeof(why))) HError(7399, "StartRecognition: stream widths: %s", why);
   }
   if (hset->hsKind != PLAINHS && hset->hsKind != SHAREDHS) HError(7399, "StartRecognition: tied-mixture and discrete systems are not supported");
   if (hset->xf != NULL) HError(7399, "StartRecognition: input transforms are not supported");
   NewHMMScan(hset, &hss);
   do {
      HLink hmm = hss.hmm;
      hmmOf[h] = hmm; pm_put(&S.hmmIdx, hmm, h);
      if (pm_get(&tm, hmm->transP) < 0) {
         if (nT + 1 > capT) { capT = capT * 2 + 64; towner = (HLink *)realloc(towner, sizeof(HLink) * (size_t)capT); }
         pm_put(&tm, hmm->transP, nT); towner[nT++] = hmm; nTp += hmm->numStates * hmm->numStates;
      }
      nHs += hmm->numStates - 2;
      for (j = 2; j < hmm->numStates; j++) {
         StateInfo *si = hmm->svec[j].info;
         if (pm_get(&sm, si) >= 0) continue;
         if (nS + 1 > capS) { capS = capS * 2 + 256; ste = (StreamElem **)realloc(ste, sizeof(StreamElem *) * (size_t)capS); }
         pm_put(&sm, si, nS); ste[nS++] = si->pdf + 1;
         sweight = (float *)realloc(sweight, sizeof(float) * (size_t)nS * NSt);
         for (ks = 0; ks < NSt; ks++) sweight[(size_t)(nS - 1) * NSt + ks] = si->weights ? si->weights[ks + 1] : 1.0f;
         for (ks = 1; ks <= NSt; ks++)
         for (k = 1; k <= si->pdf[ks].nMix; k++) {
            MixPDF *mp = si->pdf[ks].spdf.cpdf[k].mpdf;
            if (pm_get(&gm, mp) >= 0) continue;
            if (nG + 1 > capG) { capG = capG * 2 + 1024; mix = (MixPDF **)realloc(mix, sizeof(MixPDF *) * (size_t)capG); gStr = (int *)realloc(gStr, sizeof(int) * (size_t)capG); }
            pm_put(&gm, mp, nG); gStr[nG] = ks - 1; mix[nG++] = mp;
         }
      }
      h++;
   } while (GoNextHMM(&hss));
   EndHMMScan(&hss);
   stateCompOff = (int *)calloc((size_t)nS * NSt + 1, sizeof(int));           /* one entry per (state, stream) */
   for (s = 0; s < nS; s++) for (ks = 0; ks < NSt; ks++) stateCompOff[s * NSt + ks + 1] = stateCompOff[s * NSt + ks] + ste[s][ks].nMix;
   C = stateCompOff[nS * NSt];
   compGauss = (int *)calloc((size_t)C, sizeof(int)); weight = (float *)calloc((size_t)C, sizeof(float)); logwt = (float *)calloc((size_t)C, sizeof(float));
   for (s = 0; s < nS; s++)
      for (ks = 0; ks < NSt; ks++)
      for (k = 1; k <= ste[s][ks].nMix; k++) {
         MixtureElem *me = ste[s][ks].spdf.cpdf + k;
         c = stateCompOff[s * NSt + ks] + k - 1;
         if (me->mpdf->ckind != DIAGC && me->mpdf->ckind != INVDIAGC) HError(7399, "StartRecognition: only diagonal covariances are supported");
         compGauss[c] = pm_get(&gm, me->mpdf);
         if (hset->logWt) { logwt[c] = me->weight; weight[c] = (me->weight <= LMINMIX) ? 0.0f : (float)exp((double)me->weight); }
         else { weight[c] = me->weight; logwt[c] = (me->weight < MINMIX) ? (float)LZERO : (float)log((double)me->weight); }
      }
   mean = (float *)calloc((size_t)nG * S.D, sizeof(float)); var = (float *)calloc((size_t)nG * S.D, sizeof(float));
   ivar = (float *)calloc((size_t)nG * S.D, sizeof(float)); gconst = (float *)calloc((size_t)nG, sizeof(float));
   for (g = 0; g < nG; g++) {
      int kk = 0;                                          /* index in the stream's own vector */
      for (k = 1; k <= S.D; k++) {
         const size_t at = (size_t)g * S.D + k - 1;
         if (NSt > 1 && S.dimStream[k - 1] != gStr[g]) { mean[at] = 0.0f; var[at] = INFINITY; ivar[at] = 0.0f; continue; }      /* include/htk_amd.h: undivided rows */
         kk++;
         const float v = mix[g]->cov.var[kk];
         mean[at] = mix[g]->mean[kk];
         if (mix[g]->ckind == INVDIAGC) { ivar[at] = v; var[at] = 1 / v; }
         else { float c2 = v; if (c2 > 1E+30) c2 = 1E+30; if (c2 < 1E-30) c2 = 1E-30; var[at] = v; ivar[at] = 1 / c2; }
      }
      gconst[g] = mix[g]->gConst;
   }
   transN = (int *)calloc((size_t)nT, sizeof(int)); transOff = (int *)calloc((size_t)nT + 1, sizeof(int)); transP = (float *)calloc((size_t)nTp, sizeof(float));
   for (t = 0; t < nT; t++) {
      const int N = towner[t]->numStates;
      transN[t] = N; transOff[t + 1] = transOff[t] + N * N;
      for (i = 1; i <= N; i++) for (j = 1; j <= N; j++) transP[transOff[t] + (i - 1) * N + (j - 1)] = towner[t]->transP[i][j];
   }
   hmmTrans = (int *)calloc((size_t)H, sizeof(int)); hmmStateOff = (int *)calloc((size_t)H + 1, sizeof(int)); hmmState = (int *)calloc((size_t)nHs, sizeof(int));
   for (h = 0; h < H; h++) {
      hmmTrans[h] = pm_get(&tm, hmmOf[h]->transP);
      hmmStateOff[h + 1] = hmmStateOff[h] + hmmOf[h]->numStates - 2;
      for (j = 2; j < hmmOf[h]->numStates; j++) hmmState[hmmStateOff[h] + j - 2] = pm_get(&sm, hmmOf[h]->svec[j].info);
   }
   memset(&d, 0, sizeof(d));
   d.vecSize = S.D; d.numStates = nS; d.numComp = C; d.numGauss = nG; d.numTrans = nT; d.numPhys = H;
   d.stateCompOff = stateCompOff; d.compWeight = weight; d.compGauss = compGauss; d.mean = mean; d.var = var; d.gconst = gconst;
   d.transN = transN; d.transOff = transOff; d.transP = transP; d.hmmTrans = hmmTrans; d.hmmStateOff = hmmStateOff; d.hmmState = hmmState;
   d.numStreams = NSt; d.dimStream = NSt > 1 ? S.dimStream : NULL; d.streamWeight = NSt > 1 ? sweight : NULL;
   amd_check(htkamd_model_create(&d, &S.model), "htkamd_model_create");
   amd_check(htkamd_model_set_prepared(S.model, ivar, gconst, logwt), "htkamd_model_set_prepared");   /* HVite ran ConvDiagC (HVite.c:503) */
   free(weight); free(logwt); free(mean); free(var); free(ivar); free(gconst); free(transP); free(stateCompOff); free(compGauss);
   free(transN); free(transOff); free(hmmTrans); free(hmmStateOff); free(hmmState); free(hmmOf); free(ste); free(mix); free(towner); free(gStr); free(sweight);
   pm_free(&sm); pm_free(&gm); pm_free(&tm);
}

/* Network -> htkamd_net_desc.  Node 0 = net->initial, node 1 = net->final, then the chain; a node's links keep their order. */
static void pack_network(Network *net)
{
   pmap nm = {0};
   NetNode **nodes, *n;
   int nN = 2, nL = 0, i, k, at;
   int *kind, *model, *linkOff, *linkDest;
   float *pronProb, *linkLike;
   htkamd_net_desc nd;

   for (n = net->chain; n != NULL; n = n->chain) nN++;
   nodes = (NetNode **)calloc((size_t)nN, sizeof(NetNode *));
   nodes[0] = &net->initial; nodes[1] = &net->final;
   pm_put(&nm, nodes[0], 0); pm_put(&nm, nodes[1], 1);
   for (n = net->chain, i = 2; n != NULL; n = n->chain, i++) { nodes[i] = n; pm_put(&nm, n, i); }
   for (i = 0; i < nN; i++) nL += nodes[i]->nlinks;
   kind = (int *)calloc((size_t)nN, sizeof(int)); model = (int *)calloc((size_t)nN, sizeof(int)); pronProb = (float *)calloc((size_t)nN, sizeof(float));
   linkOff = (int *)calloc((size_t)nN + 1, sizeof(int)); linkDest = (int *)calloc((size_t)(nL ? nL : 1), sizeof(int)); linkLike = (float *)calloc((size_t)(nL ? nL : 1), sizeof(float));
   free(S.wordNode); S.wordNode = (NetNode **)calloc((size_t)nN, sizeof(NetNode *)); S.nWordNode = 0;
   for (i = 0, at = 0; i < nN; i++) {
      n = nodes[i];
      if (n->type & n_hmm) {
         kind[i] = HTKAMD_NODE_HMM;
         model[i] = pm_get(&S.hmmIdx, n->info.hmm);
         if (model[i] < 0) HError(7399, "StartRecognition: a network node names a model that is not in the set");
      } else if (n->info.pron != NULL) {
         kind[i] = HTKAMD_NODE_WORD; model[i] = S.nWordNode; pronProb[i] = n->info.pron->prob;
         S.wordNode[S.nWordNode++] = n;
      } else {
         if (n->tag != NULL) HError(7399, "StartRecognition: tagged null nodes (sub-lattice tags) are not supported");
         kind[i] = HTKAMD_NODE_NULL; model[i] = -1;
      }
      linkOff[i] = at;
      for (k = 0; k < n->nlinks; k++) {
         const int dst = pm_get(&nm, n->links[k].node);
         if (dst < 0) HError(7399, "StartRecognition: link to a node outside the network chain");
         linkDest[at] = dst; linkLike[at] = n->links[k].like; at++;
      }
   }
   linkOff[nN] = at;
   memset(&nd, 0, sizeof(nd));
   nd.nNodes = nN; nd.nLinks = nL; nd.nProns = S.nWordNode; nd.initial = 0; nd.final = 1;
   nd.kind = kind; nd.model = model; nd.pronProb = pronProb; nd.linkOff = linkOff; nd.linkDest = linkDest; nd.linkLike = linkLike;
   if (S.dec) { htkamd_decoder_destroy(S.dec); S.dec = NULL; }
   amd_check(htkamd_decoder_create(S.model, &nd, S.scale, &S.dec), "htkamd_decoder_create");
   free(kind); free(model); free(pronProb); free(linkOff); free(linkDest); free(linkLike); free(nodes);
   pm_free(&nm);
   S.net = net;
}

/* ------------------------------------------------------------------------------------------------ wrapped entry points */
PSetInfo *__real_InitPSetInfo(HMMSet *hset);
PSetInfo *__wrap_InitPSetInfo(HMMSet *hset)
{
   S.hset = hset;
   return __real_InitPSetInfo(hset);
}

VRecInfo *__real_InitVRecInfo(PSetInfo *psi, int nToks, Boolean models, Boolean states);
VRecInfo *__wrap_InitVRecInfo(PSetInfo *psi, int nToks, Boolean models, Boolean states)
{
   S.nToks = nToks; S.models = models; S.states = states;
   if (nToks > 8) HError(7399, "InitVRecInfo: at most 8 tokens per state (-n) on the MI355X recogniser");
   if (models || states) HError(7399, "InitVRecInfo: model / state level output (-m -f) is not served by the shim (tools/hvite does it)");
   return __real_InitVRecInfo(psi, nToks, models, states);
}

void __wrap_StartRecognition(VRecInfo *vri, Network *net, float scale, LogFloat wordpen, float pscale)
{
   if (htkamd_device_count() <= 0)
      HError(7399, "StartRecognition: %s", "no HIP device: the MI355X recogniser has no CPU path (HTKAMD_ENODEV)");
   if (S.hset == NULL) HError(7399, "StartRecognition: InitPSetInfo has not been called");
   if (S.model == NULL) pack_model();
   if (net != S.net || scale != S.scale) { S.scale = scale; pack_network(net); }
   S.wordpen = wordpen; S.pscale = pscale;
   S.nX = 0;
   vri->noTokenSurvived = TRUE;
   vri->frame = 0; vri->nact = 0;
   vri->genMaxNode = vri->wordMaxNode = NULL;
}

void __wrap_ProcessObservation(VRecInfo *vri, Observation *obs, int id, AdaptXForm *xform)
{
   int k;
   (void)id;
   if (xform != NULL) HError(7399, "ProcessObservation: input transforms are not supported");
   if (S.nX + (size_t)S.D > S.capX) { S.capX = (S.capX + (size_t)S.D) * 2 + 4096; S.X = (float *)realloc(S.X, sizeof(float) * S.capX); }
   if (S.NSt <= 1) for (k = 1; k <= S.D; k++) S.X[S.nX++] = obs->fv[1][k];
   else {                                                  /* the undivided row back from the stream vectors */
      int at[SMAX] = {0};
      for (k = 0; k < S.D; k++) { const int st = S.dimStream[k]; S.X[S.nX++] = obs->fv[st + 1][++at[st]]; }
   }
   vri->frame++;
}

Lattice *__wrap_CompleteRecognition(VRecInfo *vri, HTime frameDur, MemHeap *heap)
{
   const int T = (int)(S.nX / (size_t)(S.D > 0 ? S.D : 1)), maxWords = 8192;
   void *dX = NULL;
   int frameOff[2], nW = 0, w;
   int *wPron = (int *)malloc(sizeof(int) * maxWords), *wStart = (int *)malloc(sizeof(int) * maxWords), *wEnd = (int *)malloc(sizeof(int) * maxWords);
   float *wScore = (float *)malloc(sizeof(float) * maxWords), *wLm = (float *)malloc(sizeof(float) * maxWords), *wAc = (float *)malloc(sizeof(float) * maxWords);
   double *wLike = (double *)malloc(sizeof(double) * maxWords), total = LZERO;
   float finalLm = 0.0f;
   htkamd_decode_config cfg;
   htkamd_decode_out out;
   Lattice *lat = NULL;

   vri->frameDur = frameDur;
   vri->noTokenSurvived = TRUE;
   if (T <= 0) goto done;
   amd_check(htkamd_dev_malloc(&dX, sizeof(float) * S.nX), "htkamd_dev_malloc");
   amd_check(htkamd_memcpy_h2d(dX, S.X, sizeof(float) * S.nX, NULL), "htkamd_memcpy_h2d");
   frameOff[0] = 0; frameOff[1] = T;
   memset(&cfg, 0, sizeof(cfg));
   cfg.genBeam = vri->genBeam; cfg.wordBeam = vri->wordBeam;           /* SetPruningLevels (HRec.c): -LZERO = off */
   if (!(cfg.genBeam > 0) || cfg.genBeam > 1.0e10f) cfg.genBeam = 1.0e10f;
   if (!(cfg.wordBeam > 0) || cfg.wordBeam > 1.0e10f) cfg.wordBeam = 1.0e10f;
   cfg.lmScale = S.scale; cfg.wordPen = S.wordpen; cfg.prScale = S.pscale; cfg.scoreMode = HTKAMD_SCORE_EXACT;
   cfg.maxActive = vri->maxBeam > 0 ? vri->maxBeam : 0;      /* HVite -u */
   if (S.nToks > 1) {
      /* ---- N-best (HVite -n): token sets on the device, then the Lattice CreateLattice would have built (HRec.c:1679): node i / arc j
         of htkamd_decoder_run_lattice become lnodes[i] / larcs[j] (node 0 = start, node 1 = end); the reference's own WriteLattice and
         TranscriptionFromLattice take it from there */
      const int maxN = 65536, maxA = 262144;
      int nn = 0, na = 0, i;
      int *nodeFrame = (int *)malloc(sizeof(int) * maxN), *nodePron = (int *)malloc(sizeof(int) * maxN), *aS = (int *)malloc(sizeof(int) * maxA), *aE = (int *)malloc(sizeof(int) * maxA);
      double *nodeLike = (double *)malloc(sizeof(double) * maxN), *aSc = (double *)malloc(sizeof(double) * maxA);
      float *aAc = (float *)malloc(sizeof(float) * maxA), *aLm = (float *)malloc(sizeof(float) * maxA), *aPr = (float *)malloc(sizeof(float) * maxA);
      htkamd_lattice_out lo;
      float nBeam = vri->nBeam;
      if (!(nBeam > 0) || nBeam > 1.0e10f) nBeam = 1.0e10f;
      memset(&lo, 0, sizeof(lo));
      lo.nNodes = &nn; lo.nArcs = &na; lo.nodeFrame = nodeFrame; lo.nodePron = nodePron; lo.nodeLike = nodeLike;
      lo.arcStart = aS; lo.arcEnd = aE; lo.arcAc = aAc; lo.arcLm = aLm; lo.arcPr = aPr; lo.arcScore = aSc; lo.total = &total;
      amd_check(htkamd_decoder_run_lattice(S.dec, &cfg, S.nToks, nBeam, (const float *)dX, frameOff, 1, maxN, maxA, &lo, NULL), "htkamd_decoder_run_lattice");
      amd_check(htkamd_dev_free(dX), "htkamd_dev_free");
      if (nn == -3) HError(7399, "CompleteRecognition: the lattice has more than %d nodes / %d arcs", maxN, maxA);
      if (nn > 0) {
         vri->noTokenSurvived = FALSE;
         lat = NewLattice(heap, nn, na);
         lat->voc = S.net->vocab;
         lat->lmscale = S.scale; lat->wdpenalty = S.wordpen; lat->prscale = S.pscale; lat->framedur = frameDur;
         for (i = 0; i < nn; i++) {
            LNode *ln = lat->lnodes + i;
            ln->time = nodeFrame[i] * frameDur; ln->tag = NULL; ln->score = nodeLike[i];
            if (nodePron[i] >= 0) { NetNode *wn = S.wordNode[nodePron[i]]; ln->word = wn->info.pron->word; ln->tag = wn->tag; ln->v = wn->info.pron->pnum; }
            else { ln->word = NULL; ln->v = (i == 0) ? -1 : 0; }
         }
         for (i = 0; i < na; i++) {
            LArc *la = lat->larcs + i;
            la->start = lat->lnodes + aS[i]; la->end = lat->lnodes + aE[i];
            la->aclike = aAc[i]; la->lmlike = aLm[i]; la->prlike = aPr[i]; la->score = aSc[i];
            la->farc = la->start->foll; la->parc = la->end->pred; la->start->foll = la->end->pred = la;
         }
      }
      free(nodeFrame); free(nodePron); free(aS); free(aE); free(nodeLike); free(aSc); free(aAc); free(aLm); free(aPr);
      goto done;
   }
   memset(&out, 0, sizeof(out));
   out.nWords = &nW; out.wordPron = wPron; out.wordStart = wStart; out.wordEnd = wEnd; out.wordScore = wScore; out.wordLm = wLm; out.wordAc = wAc;
   out.wordLike = wLike; out.total = &total; out.finalLm = &finalLm;
   amd_check(htkamd_decoder_run_out(S.dec, &cfg, (const float *)dX, frameOff, 1, maxWords, &out, NULL), "htkamd_decoder_run_out");
   amd_check(htkamd_dev_free(dX), "htkamd_dev_free");
   if (nW == -3) HError(7399, "CompleteRecognition: more than %d words in the best path", maxWords);
   if (nW < 0) goto done;                               /* no token reached the end of the network */
   vri->noTokenSurvived = FALSE;
   /* CreateLattice / LatFromPaths for the single best path: nW + 2 nodes, nW + 1 arcs, numbered from the end backwards */
   lat = NewLattice(heap, nW + 2, nW + 1);
   lat->voc = S.net->vocab;
   lat->lmscale = S.scale; lat->wdpenalty = S.wordpen; lat->prscale = S.pscale; lat->framedur = frameDur;
   lat->lnodes[0].time = 0.0; lat->lnodes[0].word = NULL; lat->lnodes[0].tag = NULL; lat->lnodes[0].score = 0.0;
   {
      Word nullWord = GetWord(lat->voc, GetLabId("!NULL", FALSE), FALSE);
      LNode *endN = lat->lnodes + 1;
      int ln = 0;
      LArc *la;
      (void)nullWord;
      endN->time = T * frameDur; endN->word = NULL; endN->tag = NULL; endN->v = 0; endN->score = total;
      /* arc 0: last word end -> end of utterance (the dummy path on top of the final token: no word, no penalty, lm of the token) */
      la = lat->larcs + ln++;
      la->start = (nW > 0) ? lat->lnodes + 2 : lat->lnodes; la->end = endN;
      {
         const double prlk = (nW > 0) ? wLike[nW - 1] : 0.0;
         const float lmTok = finalLm;                     /* what the final token collected after the last word end */
         la->aclike = (float)(total - prlk - lmTok * S.scale - 0.0);
         la->prlike = 0.0; la->lmlike = lmTok; la->score = total;
      }
      la->farc = la->start->foll; la->parc = la->end->pred; la->start->foll = la->end->pred = la;
      for (w = nW - 1; w >= 0; w--) {                    /* node 2 + (nW-1-w) = end of word w */
         LNode *ne = lat->lnodes + 2 + (nW - 1 - w), *ns = (w > 0) ? lat->lnodes + 2 + (nW - w) : lat->lnodes;
         NetNode *wn = S.wordNode[wPron[w]];
         ne->time = wEnd[w] * frameDur; ne->word = wn->info.pron->word; ne->tag = wn->tag; ne->v = wn->info.pron->pnum; ne->score = wLike[w];
         la = lat->larcs + ln++;
         la->start = ns; la->end = ne;
         la->aclike = wAc[w]; la->prlike = wn->info.pron->prob; la->lmlike = wLm[w]; la->score = wLike[w];
         la->farc = ns->foll; la->parc = ne->pred; ns->foll = ne->pred = la;
      }
   }
done:
   free(wPron); free(wStart); free(wEnd); free(wScore); free(wLm); free(wAc); free(wLike);
   S.nX = 0;
   return lat;
}
