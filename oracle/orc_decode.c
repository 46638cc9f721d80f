/* orc_decode.c -- CPU restatement of HRec's 1-best token passing over a recognition network (TEST INFRASTRUCTURE).
 *
 * The network is the flat form of include/htk_amd.h (htkamd_net_desc): HMM / WORD / NULL nodes, links with LM log
 * probabilities, node `initial` and node `final`.  Semantics follow HRec.c with nToks = 1 and no alignment records:
 *   StartRecognition (HRec.c:1884): thresholds at LSMALL, token (like 0, lm 0, no path) in the initial node, one pass 2.
 *   ProcessObservation (HRec.c:1935-2030) per frame:
 *     pass 1  StepHMM1 (:642) on every HMM instance -- best predecessor per state over seIndex (first maximum wins),
 *             output probability added if the best is above the PREVIOUS frame's genThresh, entry token consumed,
 *             exit token = best of like_i + a_iN (tee transition excluded), instance max, genMaxTok; the word-end
 *             beam's top wordMaxTok = max(exit + LikeToWord(node)) over nodes with a zero-time link to a word (:758-763,
 *             LikeToWord :1172).  StepWord1 (:1038) empties word/null nodes.
 *     thresholds (:1997-2004): float(genMax - genBeam), float(wordMax - wordBeam), floored at LSMALL.
 *     pass 2  instances whose max is under genThresh are detached; StepInst2 (:1360): WORD nodes add wordpen +
 *             pron prob * pscale and open a Path record (StepWord2 :1046), NULL nodes pass the token, tee models pass
 *             entry -> exit (StepHMM2 :790); a word token under wordThresh dies; a token above genThresh goes down every
 *             link with like += lm*scale (float product), lm += lm, into SetEntryState (:1303: strict >).
 *             The reference keeps its instance list in propagation order (ReOrderList :1152) so that every zero-time
 *             node is stepped after its predecessors; here the zero-time nodes are walked in a topological order.
 *   CompleteRecognition (:2054): the final node's exit token; CreateLattice/LatFromPaths (:1512) turn its Path chain
 *     into arcs (aclike, lmlike, prlike as floats) and TranscriptionFromLattice (:2176) labels every real word with
 *     LArcTotLike (HNet.h:257).
 * Ties between tokens of exactly equal likelihood are resolved by list order in the reference and by node order here.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "htk_oracle.h"
#include "orc_ilist.h"

typedef struct { double like; float lm; int path; } tok_t;
typedef struct { int prev, node, frame; double like; float lm; } path_t;

static const tok_t NULLTOK = { ORC_LZERO, 0.0f, -1 };

#define KIND_HMM 0
#define KIND_WORD 1
#define KIND_NULL 2

typedef struct {
   const orc_model *m;
   int nNodes; const int *kind, *model; const float *pronProb; const int *linkOff, *linkDest; const float *linkLike;
   int *N, *tok0, *tee; const float **tp; int *seLo, *seHi;       /* per HMM node */
   float *wdlk;
} dnet;

#define TPN(d,n,i,j) ((d)->tp[n][((i)-1)*(d)->N[n] + ((j)-1)])

static int zero_time(const dnet *d, int n) { return d->kind[n] != KIND_HMM || d->tee[n]; }

/* LikeToWord (HRec.c:1172): best LM look-ahead to a word node over zero-time links */
static float like_to_word(const dnet *d, int n, float scale)
{
   float best = (float)ORC_LZERO;
   for (int k = d->linkOff[n]; k < d->linkOff[n + 1]; k++) {
      const int dst = d->linkDest[k];
      if (!zero_time(d, dst)) continue;
      float like = d->linkLike[k] * scale;
      if (like <= best) continue;
      if (d->kind[dst] != KIND_HMM) { if (like > best) best = like; }
      else {                                                  /* tee model on the way to the word */
         like += TPN(d, dst, 1, d->N[dst]);
         like += like_to_word(d, dst, scale);
         if (like > best) best = like;
      }
   }
   return best;
}

/* IsWd0Link (HNet.c:1663): the link leads to a word node, directly or through tee models */
static int is_wd0_link(const dnet *d, int dst)
{
   if (d->kind[dst] != KIND_HMM) return 1;
   if (!d->tee[dst]) return 0;
   for (int k = d->linkOff[dst]; k < d->linkOff[dst + 1]; k++) if (is_wd0_link(d, d->linkDest[k])) return 1;
   return 0;
}

static int cmp_desc(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return (x < y) - (x > y); }

int orc_decode(const orc_model *m, const float *X, int T,
               int nNodes, const int *kind, const int *model, const float *pronProb,
               const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
               float genBeam, float wordBeam, float lmScale, float wordPen, float prScale,
               int maxWords, int *wordPron, int *wordStart, int *wordEnd, float *wordScore, double *totalLike)
{
   return orc_decode_u(m, X, T, nNodes, kind, model, pronProb, linkOff, linkDest, linkLike, initial, final, genBeam, wordBeam, lmScale, wordPen, prScale,
                       0, maxWords, wordPron, wordStart, wordEnd, wordScore, totalLike);
}

/* the same with HVite -u: maximum-model pruning (ProcessObservation HRec.c:1966-1985) -- before pass 1 of a frame, when more than
   maxActive instances are attached, the instances whose max lies below the (maxActive+1)-th largest max (as floats) are detached */
int orc_decode_u(const orc_model *m, const float *X, int T,
               int nNodes, const int *kind, const int *model, const float *pronProb,
               const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
               float genBeam, float wordBeam, float lmScale, float wordPen, float prScale, int maxActive,
               int maxWords, int *wordPron, int *wordStart, int *wordEnd, float *wordScore, double *totalLike)
{
   dnet d; memset(&d, 0, sizeof(d));
   d.m = m; d.nNodes = nNodes; d.kind = kind; d.model = model; d.pronProb = pronProb; d.linkOff = linkOff; d.linkDest = linkDest; d.linkLike = linkLike;
   d.N = (int *)calloc((size_t)nNodes, sizeof(int)); d.tok0 = (int *)calloc((size_t)nNodes + 1, sizeof(int)); d.tee = (int *)calloc((size_t)nNodes, sizeof(int));
   d.tp = (const float **)calloc((size_t)nNodes, sizeof(float *));
   d.wdlk = (float *)malloc(sizeof(float) * (size_t)nNodes);
   int nTok = 0, maxN = 1, n, i, j, t, rc = -1;
   for (n = 0; n < nNodes; n++) {
      d.tok0[n] = nTok;
      if (kind[n] == KIND_HMM) {
         const int ti = m->hmmTrans[model[n]];
         d.N[n] = m->transN[ti]; d.tp[n] = m->transP + m->transOff[ti];
         d.tee[n] = TPN(&d, n, 1, d.N[n]) > ORC_LSMALL;
         nTok += d.N[n] - 1;                                  /* states 1..N-1 */
         if (d.N[n] > maxN) maxN = d.N[n];
      } else { d.N[n] = 2; nTok += 1; }
   }
   d.tok0[nNodes] = nTok;
   d.seLo = (int *)calloc((size_t)nNodes * (maxN + 1), sizeof(int)); d.seHi = (int *)calloc((size_t)nNodes * (maxN + 1), sizeof(int));
   for (n = 0; n < nNodes; n++) {
      if (kind[n] != KIND_HMM) continue;
      const int N = d.N[n];
      for (j = 2; j <= N; j++) {                              /* CreateSEIndex (HRec.c:1403) */
         int mn, mx;
         for (mn = (j == N) ? 2 : 1; mn < N; mn++) if (TPN(&d, n, mn, j) > ORC_LSMALL) break;
         for (mx = N - 1; mx > 1; mx--) if (TPN(&d, n, mx, j) > ORC_LSMALL) break;
         if (mn > mx) { mn = (j == N) ? 2 : 1; mx = N - 1; }
         d.seLo[n * (maxN + 1) + j] = mn; d.seHi[n * (maxN + 1) + j] = mx;
      }
   }
   /* a zero-time loop is refused (ExpandWordNet would have refused the lattice): Kahn over links whose destination is a zero-time node */
   {
      int *indeg = (int *)calloc((size_t)nNodes, sizeof(int)), *order = (int *)malloc(sizeof(int) * (size_t)nNodes), nOrd = 0, nz = 0;
      for (n = 0; n < nNodes; n++)
         if (zero_time(&d, n))
            for (int k = linkOff[n]; k < linkOff[n + 1]; k++) if (zero_time(&d, linkDest[k])) indeg[linkDest[k]]++;
      for (n = 0; n < nNodes; n++) if (zero_time(&d, n) && indeg[n] == 0) order[nOrd++] = n;
      for (i = 0; i < nOrd; i++) {
         n = order[i];
         for (int k = linkOff[n]; k < linkOff[n + 1]; k++) { const int dst = linkDest[k]; if (zero_time(&d, dst) && --indeg[dst] == 0) order[nOrd++] = dst; }
      }
      for (n = 0; n < nNodes; n++) if (zero_time(&d, n)) nz++;
      free(indeg); free(order);
      if (nz != nOrd) { rc = -4; goto done0; }
   }
   for (n = 0; n < nNodes; n++) {
      int wd0 = 0;
      if (kind[n] == KIND_HMM)
         for (int k = linkOff[n]; k < linkOff[n + 1]; k++) if (is_wd0_link(&d, linkDest[k])) wd0 = 1;   /* n_wd0 (HNet.c:3626-3631) */
      d.wdlk[n] = wd0 ? like_to_word(&d, n, lmScale) : (float)ORC_LZERO;
   }

   tok_t *tk = (tok_t *)malloc(sizeof(tok_t) * (size_t)nTok), *ex = (tok_t *)malloc(sizeof(tok_t) * (size_t)nNodes), *nw = (tok_t *)malloc(sizeof(tok_t) * (size_t)(maxN + 1));
   double *imax = (double *)malloc(sizeof(double) * (size_t)nNodes);
   float *qsa = (float *)malloc(sizeof(float) * (size_t)(nNodes + 1));
   int capP = 1024, nP = 0;
   path_t *pth = (path_t *)malloc(sizeof(path_t) * (size_t)capP);
   float genThresh = (float)ORC_LSMALL, wordThresh = (float)ORC_LSMALL;
   float *scv = (float *)malloc(sizeof(float) * (size_t)m->S);          /* cSOutP's state cache (HRec.c:438-470): one evaluation per frame */
   int *sct = (int *)calloc((size_t)m->S, sizeof(int));
   for (i = 0; i < nTok; i++) tk[i] = NULLTOK;
   for (n = 0; n < nNodes; n++) { ex[n] = NULLTOK; imax[n] = ORC_LZERO; }
   tok_t finalTok = NULLTOK;

   /* The instance list (HRec.c:1123-1270), restated with the node as the instance (a node has at most one): a doubly linked list
      between two sentinels in which AttachInst appends, MoveToRecent re-appends and DetachInst unlinks; pass 2 walks it from the head
      while it changes.  Its ORDER decides which of two exactly equal tokens reaches a node first (SetEntryState keeps the first: strict >). */
   orc_ilist L;
   L.nNodes = nNodes; L.linkOff = linkOff; L.linkDest = linkDest;
   L.link = (int *)malloc(sizeof(int) * (size_t)(nNodes + 2)); L.knil = (int *)malloc(sizeof(int) * (size_t)(nNodes + 2));
   L.att = (char *)calloc((size_t)nNodes, 1); L.ooo = (char *)calloc((size_t)nNodes, 1); L.tr0 = (char *)calloc((size_t)nNodes, 1);
   for (n = 0; n < nNodes; n++) L.tr0[n] = (char)zero_time(&d, n);
   L.link[nNodes] = nNodes + 1; L.knil[nNodes + 1] = nNodes; L.link[nNodes + 1] = -1; L.knil[nNodes] = -1; L.nxtInst = -1;
   const int HEAD = nNodes, TAIL = nNodes + 1;

#define ATTACH(n_) do { const int a_ = (n_); for (int i_ = d.tok0[a_]; i_ < d.tok0[a_ + 1]; i_++) tk[i_] = NULLTOK; ex[a_] = NULLTOK; imax[a_] = ORC_LZERO; \
      orc_ilist_attach(&L, a_); } while (0)
#define DETACH(n_) do { const int a_ = (n_); for (int i_ = d.tok0[a_]; i_ < d.tok0[a_ + 1]; i_++) tk[i_] = NULLTOK; ex[a_] = NULLTOK; imax[a_] = ORC_LZERO; \
      orc_ilist_detach(&L, a_); } while (0)
   /* SetEntryState (HRec.c:1303): the first of equal tokens stays; NetInst.max is a LogFloat (HRec.c:138) */
#define ENTER(dst, src) do { const int e_ = (dst); if (!L.att[e_]) ATTACH(e_); tok_t *r_ = &tk[d.tok0[e_]]; \
      if ((src).like > r_->like) *r_ = (src); if (r_->like > imax[e_]) imax[e_] = (float)r_->like; } while (0)

   /* StepInst2 (HRec.c:1360) -- may run twice on a node in one frame (a node moved behind a new predecessor is stepped again) */
#define STEP_INST2(n_) do { const int s_ = (n_); tok_t *st_ = &tk[d.tok0[s_]]; \
      if (kind[s_] == KIND_WORD) {                           /* StepWord2 (HRec.c:1046): a Path record per call */ \
         tok_t e_ = *st_; e_.like += wordPen; e_.like += pronProb[s_] * prScale; \
         if (nP + 1 > capP) { capP *= 2; pth = (path_t *)realloc(pth, sizeof(path_t) * (size_t)capP); } \
         pth[nP].prev = st_->path; pth[nP].node = s_; pth[nP].frame = t; pth[nP].like = e_.like; pth[nP].lm = e_.lm; \
         e_.path = nP++; e_.lm = 0.0f; ex[s_] = e_; \
      } else if (kind[s_] == KIND_NULL) ex[s_] = *st_; \
      else if (d.tee[s_]) {                                  /* StepHMM2 (HRec.c:790) */ \
         const double c_ = st_->like + TPN(&d, s_, 1, d.N[s_]); \
         if (c_ > ex[s_].like) { ex[s_] = *st_; ex[s_].like = c_; } \
      } \
      tok_t tok_ = ex[s_]; \
      if (kind[s_] != KIND_HMM && tok_.like < wordThresh) tok_ = NULLTOK; \
      if (tok_.like > genThresh) for (int k_ = linkOff[s_]; k_ < linkOff[s_ + 1]; k_++) { \
         tok_t x_ = tok_; const float lm_ = linkLike[k_]; x_.like = tok_.like + lm_ * lmScale; x_.lm = tok_.lm + lm_; \
         if (x_.like > genThresh) ENTER(linkDest[k_], x_); } } while (0)
#define PASS2() do { int cur_, next_; for (cur_ = L.link[HEAD]; cur_ != TAIL; cur_ = next_) { \
         if (imax[cur_] < genThresh) { next_ = L.link[cur_]; DETACH(cur_); } \
         else { L.nxtInst = cur_; STEP_INST2(cur_); next_ = L.link[L.nxtInst]; } } } while (0)

   /* StartRecognition (HRec.c:1884): the initial node's instance with a token of likelihood 0, thresholds at LSMALL, one pass 2 */
   t = 0;
   ATTACH(initial);
   tk[d.tok0[initial]].like = 0.0; tk[d.tok0[initial]].lm = 0.0f; tk[d.tok0[initial]].path = -1; imax[initial] = 0.0;
   PASS2();
   if (T == 0) finalTok = L.att[final] ? ex[final] : NULLTOK;

   for (t = 1; t <= T; t++) {
      double genMax = ORC_LZERO, wordMax = ORC_LZERO;
      if (maxActive > 0) {                                    /* HRec.c:1966-1985 */
         int nact = 0;
         for (n = L.link[HEAD]; n != TAIL; n = L.link[n]) qsa[nact++] = (float)imax[n];
         if (nact > maxActive) {
            qsort(qsa, (size_t)nact, sizeof(float), cmp_desc);
            const float thresh = qsa[maxActive];
            if (thresh > ORC_LSMALL) {
               int nx;
               for (n = L.link[HEAD]; n != TAIL; n = nx) { nx = L.link[n]; if (imax[n] < thresh) DETACH(n); }
            }
         }
      }
      /* pass 1 (StepInst1 on every instance; its order decides nothing: the beams' tops are maxima) */
      for (n = L.link[HEAD]; n != TAIL; n = L.link[n]) {
         if (kind[n] != KIND_HMM) { tk[d.tok0[n]] = NULLTOK; ex[n] = NULLTOK; imax[n] = ORC_LZERO; continue; }   /* StepWord1 */
         const int N = d.N[n];
         tok_t *s = tk + d.tok0[n] - 1;                       /* s[1..N-1] */
         double mx = ORC_LZERO;
         for (j = 2; j < N; j++) {
            int arg = d.seLo[n * (maxN + 1) + j];
            tok_t best = s[arg]; best.like += TPN(&d, n, arg, j);
            for (i = arg + 1; i <= d.seHi[n * (maxN + 1) + j]; i++) {
               const double c = s[i].like + TPN(&d, n, i, j);
               if (c > best.like) { best = s[i]; best.like = c; }
            }
            if (best.like > genThresh) {
               const int st = m->hmmState[m->hmmStateOff[model[n]] + (j - 2)];
               if (sct[st] != t) { scv[st] = orc_state_outp(m, st, X + (size_t)(t - 1) * m->D, NULL); sct[st] = t; }
               best.like += scv[st];
               nw[j] = best;
               if (best.like > mx) mx = best.like;
            } else nw[j] = NULLTOK;
         }
         s[1] = NULLTOK;
         for (j = 2; j < N; j++) s[j] = nw[j];
         imax[n] = (float)mx;                                 /* inst->max = max.like: a LogFloat */
         if (mx > genMax) genMax = mx;
         {
            int arg = d.seLo[n * (maxN + 1) + N];
            tok_t best = s[arg]; best.like += TPN(&d, n, arg, N);
            for (i = arg + 1; i <= d.seHi[n * (maxN + 1) + N]; i++) {
               const double c = s[i].like + TPN(&d, n, i, N);
               if (c > best.like) { best = s[i]; best.like = c; }
            }
            if (best.like > ORC_LSMALL) {
               ex[n] = best;
               const double w = best.like + d.wdlk[n];
               if (w > wordMax) wordMax = w;
            } else ex[n] = NULLTOK;
         }
      }
      wordThresh = (float)(wordMax - wordBeam); if (wordThresh < ORC_LSMALL) wordThresh = (float)ORC_LSMALL;
      genThresh = (float)(genMax - genBeam); if (genThresh < ORC_LSMALL) genThresh = (float)ORC_LSMALL;
      /* pass 2 (HRec.c:2011-2021): the list from its head -- an instance under the beam is detached, the others are stepped; what a step
         attaches or moves lies behind the cursor and is reached in the same walk */
      PASS2();
      if (t == T) finalTok = L.att[final] ? ex[final] : NULLTOK;
   }
   free(L.link); free(L.knil); free(L.att); free(L.ooo); free(L.tr0);

   *totalLike = ORC_LZERO;
   rc = -1;
   if (finalTok.path >= 0) {
      /* CreateLattice: a dummy end path on top of the final token, then one arc per path record */
      int nW = 0, p;
      *totalLike = finalTok.like;
      for (p = finalTok.path; p >= 0; p = pth[p].prev) if (kind[pth[p].node] == KIND_WORD) nW++;
      if (nW > maxWords) rc = -3;
      else {
         int w = nW;
         for (p = finalTok.path; p >= 0; p = pth[p].prev) {
            if (kind[pth[p].node] != KIND_WORD) continue;
            const int prev = pth[p].prev;
            const double prlk = (prev >= 0) ? pth[prev].like : 0.0;
            const double wp = wordPen;
            float aclike = (float)(pth[p].like - prlk - pth[p].lm * lmScale - wp);
            aclike -= pronProb[pth[p].node] * prScale;
            const float lmlike = pth[p].lm, prlike = pronProb[pth[p].node];
            const float sc = (float)((double)((aclike * 1.0f + lmlike * lmScale) + prlike * prScale) + (double)wordPen);
            w--;
            wordPron[w] = model[pth[p].node]; wordEnd[w] = pth[p].frame; wordStart[w] = (prev >= 0) ? pth[prev].frame : 0; wordScore[w] = sc;
         }
         rc = nW;
      }
   }
   free(tk); free(ex); free(nw); free(imax); free(pth); free(scv); free(sct); free(qsa);
done0:
   free(d.N); free(d.tok0); free(d.tee); free(d.tp); free(d.wdlk); free(d.seLo); free(d.seHi);
   return rc;
}
