/* orc_decode_n.c -- CPU restatement of HRec's N-BEST token passing and lattice generation (TEST INFRASTRUCTURE).
 *
 * Same network form, passes and thresholds as orc_decode.c (HRec.c with nToks = 1); what is added is HVite -n N:
 *   TokenSet (HRec.c:100-108): the most likely token of a state plus up to N-1 RelTokens {like relative to it (float), lm, path},
 *       kept sorted by likelihood, DISTINCT in the word-end node their path ends in (the word-pair approximation);
 *   TokSetMerge (:279-425), restated operand for operand: exchange when the newcomer wins, match on the last word node, replace
 *       the least likely entry when the set is full, the nThresh cut (nThresh = genMax - nBeam, :2002; HVite sets nBeam = genBeam);
 *   StepHMM1 (:642) / StepHMM2 (:790) / SetEntryState (:1303) merging sets where the 1-best code takes a maximum;
 *   StepWord2 (:1046): a Path record for the best token with one NxtPath per alternative (like = path like + relative like);
 *   CompleteRecognition (:2054) -> CreateLattice (:1679): MarkPaths (:1664) numbers the Path records reachable from the final
 *       token set depth-first (the lattice nodes; node 0 = start, node 1 = the end), LatFromPaths (:1512) turns every Path and every
 *       NxtPath into an arc with  aclike = like - like(prev) - lm*scale - wordpen - pronprob*pscale,  lmlike = lm,  prlike = pron prob.
 * Null nodes (no pronunciation) make no Path records here either (the networks of net.c carry no tags).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "htk_oracle.h"
#include "orc_ilist.h"

#define MAXTOK 16
typedef struct { double like; float lm; int path; int align; } tok_t;             /* align: Token.align (HRec.h), an index into dnet.al, -1 = NULL */
typedef struct { float like, lm; int path; int align; } rtok_t;                   /* RelToken.align exists with -DPHNALG (HTKLib/Makefile.in:45) */
typedef struct { int node, state, frame, prev; double like; } align_t;            /* Align (HRec.c:150-164) */
typedef struct { tok_t tok; int n; rtok_t set[MAXTOK]; } tset_t;
typedef struct { int prev, node, frame; double like; float lm; int chain0, nChain; int usage; int align; } path_t;
typedef struct { int prev; double like; float lm; int align; } nxt_t;

static const tok_t NULLTOK = { ORC_LZERO, 0.0f, -1, -1 };

#define KIND_HMM 0
#define KIND_WORD 1
#define KIND_NULL 2

typedef struct {
   const orc_model *m;
   int nNodes; const int *kind, *model; const float *pronProb; const int *linkOff, *linkDest; const float *linkLike;
   int *N, *tok0, *tee; const float **tp; int *seLo, *seHi;
   float *wdlk;
   /* N-best state */
   int nToks; float nThresh;
   path_t *pth; int nP, capP;
   nxt_t *nxt; int nX, capX;
   int *aux;                       /* per network node: NetNode.aux of TokSetMerge */
   /* alignment records (HVite -m / -f together with -n: pri->models / pri->states) */
   int models, states;
   align_t *al; int nAl, capAl;
} dnet;

/* NewNRefAlign (HRec.c:598-622) */
static int new_align(dnet *d, int node, int state, double like, int frame, int prev)
{
   if (d->nAl + 1 > d->capAl) { d->capAl = d->capAl ? d->capAl * 2 : 4096; d->al = (align_t *)realloc(d->al, sizeof(align_t) * (size_t)d->capAl); }
   align_t *a = &d->al[d->nAl];
   a->node = node; a->state = state; a->like = like; a->frame = frame; a->prev = prev;
   return d->nAl++;
}

#define TPN(d,n,i,j) ((d)->tp[n][((i)-1)*(d)->N[n] + ((j)-1)])

static int zero_time(const dnet *d, int n) { return d->kind[n] != KIND_HMM || d->tee[n]; }

static float like_to_word(const dnet *d, int n, float scale)
{
   float best = (float)ORC_LZERO;
   for (int k = d->linkOff[n]; k < d->linkOff[n + 1]; k++) {
      const int dst = d->linkDest[k];
      if (!zero_time(d, dst)) continue;
      float like = d->linkLike[k] * scale;
      if (like <= best) continue;
      if (d->kind[dst] != KIND_HMM) { if (like > best) best = like; }
      else { like += TPN(d, dst, 1, d->N[dst]); like += like_to_word(d, dst, scale); if (like > best) best = like; }
   }
   return best;
}
static int is_wd0_link(const dnet *d, int dst)
{
   if (d->kind[dst] != KIND_HMM) return 1;
   if (!d->tee[dst]) return 0;
   for (int k = d->linkOff[dst]; k < d->linkOff[dst + 1]; k++) if (is_wd0_link(d, d->linkDest[k])) return 1;
   return 0;
}

static void set_null(tset_t *s) { s->tok = NULLTOK; s->n = 1; }

/* the network node of the last real word on a path, -1 for "no word yet" (TokSetMerge's walk down path->prev past null nodes) */
static int word_node_of(const dnet *d, int path) { return path < 0 ? -1 : d->pth[path].node; }

/* TokSetMerge (HRec.c:279): token `cmp` with the relative tokens of `src` merged into `res` */
static void tokset_merge(dnet *d, tset_t *res, const tok_t *cmp, const tset_t *src)
{
   tset_t tmp;
   int i, k, nw = 0, nullIdx = 0, nodes[MAXTOK];
   float diff, like, limit;
   if (cmp->like >= res->tok.like) {
      if (cmp->like > d->nThresh) {
         if (res->tok.like > d->nThresh) {                 /* exchange res and src */
            tmp.tok = res->tok; tmp.n = res->n;
            for (k = 0; k < res->n; k++) tmp.set[k] = res->set[k];
            res->tok = *cmp;
            for (k = 0; k < src->n; k++) res->set[k] = src->set[k];
            res->n = src->n;
         } else {
            res->tok = *cmp;
            for (k = 0; k < src->n; k++) res->set[k] = src->set[k];
            res->n = src->n;
            return;
         }
      } else return;
   } else {
      if (cmp->like > d->nThresh) {
         tmp.tok = *cmp; tmp.n = src->n;
         for (k = 0; k < src->n; k++) tmp.set[k] = src->set[k];
      } else return;
   }
   diff = (float)(res->tok.like - tmp.tok.like);
   for (i = 0; i < res->n; i++) {
      const int node = word_node_of(d, res->set[i].path);
      if (node < 0) nullIdx = i + 1;
      else { d->aux[node] = i + 1; nodes[nw++] = node; }
   }
   limit = (float)(d->nThresh - tmp.tok.like);
   for (i = 0; i < tmp.n; i++) {
      const rtok_t *cur = &tmp.set[i];
      if (cur->like < limit) break;
      const int node = word_node_of(d, cur->path);
      const int aux = (node < 0) ? nullIdx : d->aux[node];
      like = cur->like - diff;
      int mch = -1;
      if (aux != 0)
         for (k = aux - 1; k < res->n; k++)
            if (word_node_of(d, res->set[k].path) == node) { mch = k; break; }
      if (mch < 0) {
         if (res->n < d->nToks) { mch = res->n++; res->set[mch].like = (float)ORC_LZERO; res->set[mch].lm = 0.0f; res->set[mch].path = -1; res->set[mch].align = -1; }
         else mch = res->n - 1;
      }
      if (like > res->set[mch].like) {
         for (mch--; mch >= 0 && like > res->set[mch].like; mch--) res->set[mch + 1] = res->set[mch];
         mch++;
         res->set[mch].path = cur->path; res->set[mch].lm = cur->lm; res->set[mch].align = cur->align; res->set[mch].like = like;
      }
   }
   for (i = 0; i < nw; i++) d->aux[nodes[i]] = 0;
}

static int cmp_desc_n(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return (x < y) - (x > y); }

int orc_decode_nbest(const orc_model *m, const float *X, int T,
                     int nNodes, const int *kind, const int *model, const float *pronProb,
                     const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
                     float genBeam, float wordBeam, float nBeam, float lmScale, float wordPen, float prScale, int nToks, int maxActive,
                     int maxLatNodes, int maxLatArcs, int *latNodeNet, int *latNodeFrame, double *latNodeLike,
                     int *latArcStart, int *latArcEnd, float *latArcAc, float *latArcLm, float *latArcPr, double *latArcScore,
                     int *nLatNodes, int *nLatArcs, double *totalLike)
{
   return orc_decode_nbest_align(m, X, T, nNodes, kind, model, pronProb, linkOff, linkDest, linkLike, initial, final, genBeam, wordBeam, nBeam, lmScale, wordPen, prScale,
                                 nToks, maxActive, 0, maxLatNodes, maxLatArcs, latNodeNet, latNodeFrame, latNodeLike, latArcStart, latArcEnd, latArcAc, latArcLm, latArcPr,
                                 latArcScore, nLatNodes, nLatArcs, totalLike, 0, NULL, NULL, NULL, NULL, NULL);
}

/* The same with alignment records (HVite -n together with -m: alignMode & 1, pri->models; -f: alignMode & 2, pri->states).  Every lattice
   arc then carries LatFromPaths' lAlign (HRec.c:1582-1656): arc j's records are [arcAlignOff[j], arcAlignOff[j+1]) of alState (the state,
   -1 for a model record) / alNode (network node of the model) / alDur (frames) / alLike (as the reference computes it, float).  -3 also
   when maxAlign records do not hold them. */
int orc_decode_nbest_align(const orc_model *m, const float *X, int T,
                     int nNodes, const int *kind, const int *model, const float *pronProb,
                     const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
                     float genBeam, float wordBeam, float nBeam, float lmScale, float wordPen, float prScale, int nToks, int maxActive, int alignMode,
                     int maxLatNodes, int maxLatArcs, int *latNodeNet, int *latNodeFrame, double *latNodeLike,
                     int *latArcStart, int *latArcEnd, float *latArcAc, float *latArcLm, float *latArcPr, double *latArcScore,
                     int *nLatNodes, int *nLatArcs, double *totalLike,
                     int maxAlign, int *arcAlignOff, int *alState, int *alNode, int *alDur, float *alLike)
{
   dnet d; memset(&d, 0, sizeof(d));
   d.models = (alignMode & 1) != 0; d.states = (alignMode & 2) != 0;
   if (nToks < 2 || nToks > MAXTOK) return -5;
   d.m = m; d.nNodes = nNodes; d.kind = kind; d.model = model; d.pronProb = pronProb; d.linkOff = linkOff; d.linkDest = linkDest; d.linkLike = linkLike;
   d.nToks = nToks; d.nThresh = (float)ORC_LSMALL;
   d.N = (int *)calloc((size_t)nNodes, sizeof(int)); d.tok0 = (int *)calloc((size_t)nNodes + 1, sizeof(int)); d.tee = (int *)calloc((size_t)nNodes, sizeof(int));
   d.tp = (const float **)calloc((size_t)nNodes, sizeof(float *));
   d.wdlk = (float *)malloc(sizeof(float) * (size_t)nNodes);
   d.aux = (int *)calloc((size_t)nNodes, sizeof(int));
   d.capP = 1024; d.pth = (path_t *)malloc(sizeof(path_t) * (size_t)d.capP);
   d.capX = 1024; d.nxt = (nxt_t *)malloc(sizeof(nxt_t) * (size_t)d.capX);
   int nTok = 0, maxN = 1, n, i, j, k, t, rc = -1;
   for (n = 0; n < nNodes; n++) {
      d.tok0[n] = nTok;
      if (kind[n] == KIND_HMM) {
         const int ti = m->hmmTrans[model[n]];
         d.N[n] = m->transN[ti]; d.tp[n] = m->transP + m->transOff[ti];
         d.tee[n] = TPN(&d, n, 1, d.N[n]) > ORC_LSMALL;
         nTok += d.N[n] - 1;
         if (d.N[n] > maxN) maxN = d.N[n];
      } else { d.N[n] = 2; nTok += 1; }
   }
   d.tok0[nNodes] = nTok;
   d.seLo = (int *)calloc((size_t)nNodes * (maxN + 1), sizeof(int)); d.seHi = (int *)calloc((size_t)nNodes * (maxN + 1), sizeof(int));
   for (n = 0; n < nNodes; n++) {
      if (kind[n] != KIND_HMM) continue;
      const int N = d.N[n];
      for (j = 2; j <= N; j++) {
         int mn, mx;
         for (mn = (j == N) ? 2 : 1; mn < N; mn++) if (TPN(&d, n, mn, j) > ORC_LSMALL) break;
         for (mx = N - 1; mx > 1; mx--) if (TPN(&d, n, mx, j) > ORC_LSMALL) break;
         if (mn > mx) { mn = (j == N) ? 2 : 1; mx = N - 1; }
         d.seLo[n * (maxN + 1) + j] = mn; d.seHi[n * (maxN + 1) + j] = mx;
      }
   }
   {  /* a zero-time loop is refused, as in orc_decode.c */
      int *indeg = (int *)calloc((size_t)nNodes, sizeof(int)), *order = (int *)malloc(sizeof(int) * (size_t)nNodes), nOrd = 0, nz = 0;
      for (n = 0; n < nNodes; n++)
         if (zero_time(&d, n))
            for (k = linkOff[n]; k < linkOff[n + 1]; k++) if (zero_time(&d, linkDest[k])) indeg[linkDest[k]]++;
      for (n = 0; n < nNodes; n++) if (zero_time(&d, n) && indeg[n] == 0) order[nOrd++] = n;
      for (i = 0; i < nOrd; i++) {
         n = order[i];
         for (k = linkOff[n]; k < linkOff[n + 1]; k++) { const int dst = linkDest[k]; if (zero_time(&d, dst) && --indeg[dst] == 0) order[nOrd++] = dst; }
      }
      for (n = 0; n < nNodes; n++) if (zero_time(&d, n)) nz++;
      free(indeg); free(order);
      if (nz != nOrd) { rc = -4; goto done0; }
   }
   for (n = 0; n < nNodes; n++) {
      int wd0 = 0;
      if (kind[n] == KIND_HMM)
         for (k = linkOff[n]; k < linkOff[n + 1]; k++) if (is_wd0_link(&d, linkDest[k])) wd0 = 1;
      d.wdlk[n] = wd0 ? like_to_word(&d, n, lmScale) : (float)ORC_LZERO;
   }

   tset_t *tk = (tset_t *)malloc(sizeof(tset_t) * (size_t)nTok), *ex = (tset_t *)malloc(sizeof(tset_t) * (size_t)nNodes), *nw = (tset_t *)malloc(sizeof(tset_t) * (size_t)(maxN + 1));
   double *imax = (double *)malloc(sizeof(double) * (size_t)nNodes);
   float *qsa = (float *)malloc(sizeof(float) * (size_t)(nNodes + 1));
   float genThresh = (float)ORC_LSMALL, wordThresh = (float)ORC_LSMALL;
   float *scv = (float *)malloc(sizeof(float) * (size_t)m->S);
   int *sct = (int *)calloc((size_t)m->S, sizeof(int));
   static const rtok_t RMAX = { 0.0f, 0.0f, -1, -1 };
   for (i = 0; i < nTok; i++) { set_null(&tk[i]); tk[i].set[0] = RMAX; }
   for (n = 0; n < nNodes; n++) { set_null(&ex[n]); ex[n].set[0] = RMAX; imax[n] = ORC_LZERO; }
   tset_t finalSet; set_null(&finalSet);

   /* the instance list in the reference's order (orc_ilist.h): in N-best mode it decides more than ties -- every TokSetMerge re-bases
      its relative tokens as floats (HRec.c:361-364), so the order in which the senders of a node are stepped is in the last bit of
      every alternative's likelihood */
   orc_ilist L;
   L.nNodes = nNodes; L.linkOff = linkOff; L.linkDest = linkDest;
   L.link = (int *)malloc(sizeof(int) * (size_t)(nNodes + 2)); L.knil = (int *)malloc(sizeof(int) * (size_t)(nNodes + 2));
   L.att = (char *)calloc((size_t)nNodes, 1); L.ooo = (char *)calloc((size_t)nNodes, 1); L.tr0 = (char *)calloc((size_t)nNodes, 1);
   for (n = 0; n < nNodes; n++) L.tr0[n] = (char)zero_time(&d, n);
   L.link[nNodes] = nNodes + 1; L.knil[nNodes + 1] = nNodes; L.link[nNodes + 1] = -1; L.knil[nNodes] = -1; L.nxtInst = -1;
   const int HEAD = nNodes, TAIL = nNodes + 1;

   /* AttachInst (HRec.c:1200): every set empty with one relative token rmax; DetachInst (:1270) */
#define ATTACH(n_) do { const int a_ = (n_); for (int i_ = d.tok0[a_]; i_ < d.tok0[a_ + 1]; i_++) { set_null(&tk[i_]); tk[i_].set[0] = RMAX; } \
      set_null(&ex[a_]); ex[a_].set[0] = RMAX; imax[a_] = ORC_LZERO; orc_ilist_attach(&L, a_); } while (0)
#define DETACH(n_) do { const int a_ = (n_); for (int i_ = d.tok0[a_]; i_ < d.tok0[a_ + 1]; i_++) set_null(&tk[i_]); \
      set_null(&ex[a_]); imax[a_] = ORC_LZERO; orc_ilist_detach(&L, a_); } while (0)
#define ENTER(dst, srcset) do { const int e_ = (dst); if (!L.att[e_]) ATTACH(e_); tset_t *r_ = &tk[d.tok0[e_]]; \
      tokset_merge(&d, r_, &(srcset).tok, &(srcset)); if (r_->tok.like > imax[e_]) imax[e_] = (float)r_->tok.like; } while (0)

   for (t = 0; t <= T; t++) {
      if (t == 0) {                                           /* StartRecognition (HRec.c:1884) */
         ATTACH(initial);
         tset_t *s0 = &tk[d.tok0[initial]];
         s0->tok.like = 0.0; s0->tok.lm = 0.0f; s0->tok.path = -1; s0->n = 1;
         imax[initial] = 0.0;
      } else {
         double genMax = ORC_LZERO, wordMax = ORC_LZERO;
         if (maxActive > 0) {                                 /* maximum-model pruning, HRec.c:1966-1985, as in orc_decode.c */
            int nact = 0, nx;
            for (n = L.link[HEAD]; n != TAIL; n = L.link[n]) qsa[nact++] = (float)imax[n];
            if (nact > maxActive) {
               qsort(qsa, (size_t)nact, sizeof(float), cmp_desc_n);
               const float thresh = qsa[maxActive];
               if (thresh > ORC_LSMALL)
                  for (n = L.link[HEAD]; n != TAIL; n = nx) { nx = L.link[n]; if (imax[n] < thresh) DETACH(n); }
            }
         }
         for (n = L.link[HEAD]; n != TAIL; n = L.link[n]) {    /* pass 1 */
            if (kind[n] != KIND_HMM) { set_null(&tk[d.tok0[n]]); set_null(&ex[n]); imax[n] = ORC_LZERO; continue; }   /* StepWord1 */
            const int N = d.N[n];
            tset_t *s = tk + d.tok0[n] - 1;                   /* s[1..N-1] */
            double mx = ORC_LZERO;
            for (j = 2; j < N; j++) {
               int a0 = d.seLo[n * (maxN + 1) + j];
               tset_t res = s[a0];
               res.tok.like += TPN(&d, n, a0, j);
               for (i = a0 + 1; i <= d.seHi[n * (maxN + 1) + j]; i++) {
                  tok_t c = s[i].tok; c.like += TPN(&d, n, i, j);
                  tokset_merge(&d, &res, &c, &s[i]);
               }
               if (res.tok.like > genThresh) {
                  const int st = m->hmmState[m->hmmStateOff[model[n]] + (j - 2)];
                  if (sct[st] != t) { scv[st] = orc_state_outp(m, st, X + (size_t)(t - 1) * m->D, NULL); sct[st] = t; }
                  res.tok.like += scv[st];
                  if (d.states) {                              /* HRec.c:680-704 (with -DPHNALG): a record where a token enters state j */
                     const double alk = res.tok.like - scv[st] - res.tok.lm * lmScale;
                     if (res.tok.align < 0 || d.al[res.tok.align].state != j || d.al[res.tok.align].node != n) {
                        res.tok.align = new_align(&d, n, j, alk, t - 1, res.tok.align);
                        res.set[0].align = res.tok.align;
                     }
                     for (int q = 1; q < res.n; q++)
                        if (res.set[q].align < 0 || d.al[res.set[q].align].state != j || d.al[res.set[q].align].node != n)
                           res.set[q].align = new_align(&d, n, j, alk, t - 1, res.set[q].align);      /* (the BEST token's likelihood: what the reference writes) */
                  }
                  nw[j] = res;
                  if (res.tok.like > mx) mx = res.tok.like;
               } else { nw[j] = res; set_null(&nw[j]); }
            }
            set_null(&s[1]);
            for (j = 2; j < N; j++) s[j] = nw[j];
            imax[n] = (float)mx;
            if (mx > genMax) genMax = mx;
            {
               int a0 = d.seLo[n * (maxN + 1) + N];
               tset_t res = s[a0];
               res.tok.like += TPN(&d, n, a0, N);
               for (i = a0 + 1; i <= d.seHi[n * (maxN + 1) + N]; i++) {
                  tok_t c = s[i].tok; c.like += TPN(&d, n, i, N);
                  tokset_merge(&d, &res, &c, &s[i]);
               }
               if (res.tok.like > ORC_LSMALL) {
                  const double w = res.tok.like + d.wdlk[n];
                  if (w > wordMax) wordMax = w;
                  if (!d.tee[n] && d.models) {                 /* HRec.c:762-776: the model's exit record, one per token of the set */
                     const double alk = res.tok.like - res.tok.lm * lmScale;
                     res.tok.align = new_align(&d, n, -1, alk, t, res.tok.align);
                     res.set[0].align = res.tok.align;
                     for (int q = 1; q < res.n; q++) res.set[q].align = new_align(&d, n, -1, alk, t, res.set[q].align);
                  }
                  ex[n] = res;
               } else { ex[n] = res; set_null(&ex[n]); }
            }
         }
         wordThresh = (float)(wordMax - wordBeam); if (wordThresh < ORC_LSMALL) wordThresh = (float)ORC_LSMALL;
         genThresh = (float)(genMax - genBeam); if (genThresh < ORC_LSMALL) genThresh = (float)ORC_LSMALL;
         d.nThresh = (float)(genMax - nBeam); if (d.nThresh < ORC_LSMALL / 2) d.nThresh = (float)(ORC_LSMALL / 2);
      }
      /* pass 2 (HRec.c:2011-2021; at t = 0 StartRecognition's): the list from its head while it changes */
      int cur, next;
      for (cur = L.link[HEAD]; cur != TAIL; cur = next) {
         if (imax[cur] < genThresh) { next = L.link[cur]; DETACH(cur); continue; }
         L.nxtInst = cur;
         n = cur;
         tset_t *st = &tk[d.tok0[n]];
         if (kind[n] == KIND_WORD) {                          /* StepWord2 (HRec.c:1046): a Path record, one NxtPath per alternative, per call */
            tok_t e = st->tok;
            e.like += wordPen;
            e.like += pronProb[n] * prScale;
            if (d.nP + 1 > d.capP) { d.capP *= 2; d.pth = (path_t *)realloc(d.pth, sizeof(path_t) * (size_t)d.capP); }
            path_t *np = &d.pth[d.nP];
            np->prev = st->tok.path; np->node = n; np->frame = t; np->like = e.like; np->lm = e.lm; np->usage = 0;
            np->align = e.align;
            np->chain0 = d.nX; np->nChain = 0;
            for (k = 1; k < st->n; k++) {
               if (d.nX + 1 > d.capX) { d.capX *= 2; d.nxt = (nxt_t *)realloc(d.nxt, sizeof(nxt_t) * (size_t)d.capX); }
               d.nxt[d.nX].like = np->like + st->set[k].like; d.nxt[d.nX].lm = st->set[k].lm; d.nxt[d.nX].prev = st->set[k].path;
               d.nxt[d.nX].align = st->set[k].align;
               d.nX++; np->nChain++;
            }
            e.path = d.nP++; e.lm = 0.0f; e.align = -1;
            ex[n].tok = e; ex[n].n = 1; ex[n].set[0].path = e.path;        /* set[0].like / .lm stay what AttachInst made them (rmax) */
         } else if (kind[n] == KIND_NULL) ex[n] = *st;
         else if (d.tee[n]) {                                 /* StepHMM2 (HRec.c:790) */
            tok_t c = st->tok; c.like += TPN(&d, n, 1, d.N[n]);
            tokset_merge(&d, &ex[n], &c, st);
            if (d.models) {                                  /* HRec.c:817-832 */
               tset_t *r = &ex[n];
               const double alk = r->tok.like - r->tok.lm * lmScale;
               r->tok.align = new_align(&d, n, -1, alk, t, r->tok.align);
               r->set[0].align = r->tok.align;
               for (int q = 1; q < r->n; q++) r->set[q].align = new_align(&d, n, -1, alk, t, r->set[q].align);
            }
         }
         tok_t tok = ex[n].tok;
         if (kind[n] != KIND_HMM && tok.like < wordThresh) tok = NULLTOK;
         if (tok.like > genThresh)
            for (k = linkOff[n]; k < linkOff[n + 1]; k++) {
               tset_t x; const float lm = linkLike[k];
               x.tok = ex[n].tok; x.tok.like = tok.like + lm * lmScale; x.tok.lm = tok.lm + lm; x.n = ex[n].n;
               for (int q = 0; q < x.n; q++) { x.set[q] = ex[n].set[q]; x.set[q].lm = ex[n].set[q].lm + lm; }
               if (x.tok.like > genThresh) ENTER(linkDest[k], x);
            }
         next = L.link[L.nxtInst];
      }
      if (t == T) { if (L.att[final]) finalSet = ex[final]; else set_null(&finalSet); }
   }
   free(L.link); free(L.knil); free(L.att); free(L.ooo); free(L.tr0);

   *totalLike = ORC_LZERO; *nLatNodes = 0; *nLatArcs = 0;
   rc = -1;
   if (finalSet.tok.path >= 0) {
      /* CreateLattice: a dummy end Path on top of the final token set, MarkPaths, LatFromPaths */
      *totalLike = finalSet.tok.like;
      if (d.nP + 1 > d.capP) { d.capP += 8; d.pth = (path_t *)realloc(d.pth, sizeof(path_t) * (size_t)d.capP); }
      path_t *root = &d.pth[d.nP];
      root->prev = finalSet.tok.path; root->node = -2; root->frame = T; root->like = finalSet.tok.like; root->lm = finalSet.tok.lm; root->usage = 0;
      root->align = finalSet.tok.align;
      root->chain0 = d.nX; root->nChain = 0;
      for (k = 1; k < finalSet.n; k++) {
         if (d.nX + 1 > d.capX) { d.capX *= 2; d.nxt = (nxt_t *)realloc(d.nxt, sizeof(nxt_t) * (size_t)d.capX); }
         d.nxt[d.nX].like = finalSet.tok.like + finalSet.set[k].like; d.nxt[d.nX].lm = finalSet.set[k].lm; d.nxt[d.nX].prev = finalSet.set[k].path;
         d.nxt[d.nX].align = finalSet.set[k].align;
         d.nX++; root->nChain++;
      }
      const int rootIdx = d.nP++;
      /* MarkPaths (depth first: the path itself, its best predecessor's subtree, then the alternatives in chain order) */
      int nn = 1, nl = 0, sp = 0, capS = d.nP * 2 + 16;
      int *stack = (int *)malloc(sizeof(int) * (size_t)capS);
      /* recursion unrolled with an explicit stack of (path, next child) pairs */
      int *child = (int *)calloc((size_t)d.nP, sizeof(int));
      stack[sp++] = rootIdx;
      d.pth[rootIdx].usage = -(nn++); nl++;
      while (sp > 0) {
         const int p = stack[sp - 1];
         path_t *pp = &d.pth[p];
         const int c = child[p]++;
         int nextp = -1;
         if (c == 0) nextp = pp->prev;
         else if (c - 1 < pp->nChain) { nl++; nextp = d.nxt[pp->chain0 + c - 1].prev; }
         else { sp--; continue; }
         if (nextp >= 0 && d.pth[nextp].usage >= 0) {
            d.pth[nextp].usage = -(nn++); nl++;
            if (sp + 1 > capS) { capS *= 2; stack = (int *)realloc(stack, sizeof(int) * (size_t)capS); }
            stack[sp++] = nextp;
         }
      }
      free(stack); free(child);
      if (nn > maxLatNodes || nl > maxLatArcs) rc = -3;
      else {
         int ln = 0, nAlOut = 0, alOverflow = 0;
         latNodeNet[0] = -1; latNodeFrame[0] = 0; latNodeLike[0] = 0.0;
         for (int p = 0; p < d.nP; p++) {
            const path_t *pp = &d.pth[p];
            if (pp->usage >= 0) continue;
            const int ne = -pp->usage;
            latNodeNet[ne] = pp->node; latNodeFrame[ne] = pp->frame; latNodeLike[ne] = pp->like;
            for (int c = 0; c <= pp->nChain; c++) {
               const int prev = (c == 0) ? pp->prev : d.nxt[pp->chain0 + c - 1].prev;
               const double plike = (c == 0) ? pp->like : d.nxt[pp->chain0 + c - 1].like;
               const float plm = (c == 0) ? pp->lm : d.nxt[pp->chain0 + c - 1].lm;
               const double prlk = (prev >= 0) ? d.pth[prev].like : 0.0;
               const double wp = (pp->node >= 0) ? wordPen : 0.0;
               float ac = (float)(plike - prlk - plm * lmScale - wp);
               float pr = 0.0f;
               if (pp->node >= 0) { ac -= pronProb[pp->node] * prScale; pr = pronProb[pp->node]; }
               latArcStart[ln] = (prev >= 0) ? -d.pth[prev].usage : 0; latArcEnd[ln] = ne;
               latArcAc[ln] = ac; latArcLm[ln] = plm; latArcPr[ln] = pr; latArcScore[ln] = plike;
               if (arcAlignOff) {
                  /* lAlign of this arc (HRec.c:1582-1656 with -DPHNALG): the arc's own chain of records, latest first; a state record's
                     likelihood is the running difference, a model record closes the model BEHIND it (the one met before, i.e. later) */
                  arcAlignOff[ln] = nAlOut;
                  const int a0 = (c == 0) ? pp->align : d.nxt[pp->chain0 + c - 1].align;
                  if (a0 >= 0) {
                     int cnt = 0, al;
                     for (al = a0; al >= 0; al = d.al[al].prev) cnt++;
                     if (nAlOut + cnt > maxAlign) { alOverflow = 1; }
                     else {
                        int i = cnt, frame = pp->frame, prr = -1, labpr = -1;
                        double like = plike - plm * lmScale - wp;
                        const int base = nAlOut;
                        for (al = a0; al >= 0; al = d.al[al].prev) {
                           const align_t *A = &d.al[al];
                           int durF, labNode;
                           if (A->state < 0) {
                              if (prr < 0) { prr = al; labpr = A->node; continue; }
                              durF = d.al[prr].frame - A->frame;
                              like = d.al[prr].like - A->like;
                              prr = al;
                              labNode = labpr; labpr = A->node;
                           } else {
                              labNode = A->node;
                              durF = frame - A->frame;
                              like = like - A->like;
                              frame = A->frame;
                           }
                           i--;
                           alState[base + i] = A->state; alNode[base + i] = labNode; alDur[base + i] = durF; alLike[base + i] = (float)like;
                           like = A->like;
                        }
                        if (prr >= 0) {
                           int durF;
                           if (prev >= 0) { durF = d.al[prr].frame - d.pth[prev].frame; like = d.al[prr].like - d.pth[prev].like; }
                           else { durF = d.al[prr].frame; like = d.al[prr].like; }
                           i--;
                           alState[base + i] = -1; alNode[base + i] = labpr; alDur[base + i] = durF; alLike[base + i] = (float)like;
                        }
                        nAlOut += cnt;
                     }
                  }
               }
               ln++;
            }
         }
         if (arcAlignOff) arcAlignOff[ln] = nAlOut;
         *nLatNodes = nn; *nLatArcs = ln;
         rc = alOverflow ? -3 : 0;
      }
   }
   free(tk); free(ex); free(nw); free(imax); free(qsa); free(scv); free(sct);
done0:
   free(d.N); free(d.tok0); free(d.tee); free(d.tp); free(d.wdlk); free(d.seLo); free(d.seHi); free(d.aux); free(d.pth); free(d.nxt); free(d.al);
   return rc;
}
