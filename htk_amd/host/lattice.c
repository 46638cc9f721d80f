/* lattice.c -- what HVite does with the Lattice CompleteRecognition hands back (host C over the arrays of htkamd_decoder_run_lattice):
 *   htkamd_lattice_write   WriteLattice (HNet.c:631) / WriteOneLattice (:530): the SLF text form -- header, nodes sorted by time then
 *                          likelihood (QSCmpNodes :448), arcs by end node then start node (QSCmpArcs :466), fields t W v / S E a l r
 *                          in the reference's printf formats, so that files compare byte for byte;
 *   htkamd_lattice_nbest   TranscriptionFromLattice (HRec.c:2176) for N alternatives: best completion score of every node by a
 *                          backward pass, then A* over partial paths ordered by like + completion, hypotheses with the word sequence
 *                          of an earlier answer dropped (WordMatch :2166);
 *   htkamd_lattice_arc_score   LArcTotLike (HNet.h:257), the label score.
 * Alignment records inside arcs (HVite -n with -m / -f): htkamd_lattice_write_align, OutputAlign (HNet.c:503-516).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

static const htkamd_lattice *g_lat;                /* qsort context (the reference's `slat`) */
static const int *g_rank;

static int cmp_nodes(const void *v1, const void *v2)
{
   const int s1 = *(const int *)v1, s2 = *(const int *)v2;
   const double tdiff = (double)g_lat->nodeFrame[s1] - (double)g_lat->nodeFrame[s2], sdiff = g_lat->nodeLike[s1] - g_lat->nodeLike[s2];
   if (tdiff == 0.0) { if (sdiff == 0.0) return s1 - s2; return sdiff > 0.0 ? 1 : -1; }
   return tdiff > 0.0 ? 1 : -1;
}
static int cmp_arcs(const void *v1, const void *v2)
{
   const int s1 = *(const int *)v1, s2 = *(const int *)v2;
   const int j = g_rank[g_lat->arcEnd[s1]] - g_rank[g_lat->arcEnd[s2]], k = g_rank[g_lat->arcStart[s1]] - g_rank[g_lat->arcStart[s2]];
   if (k == 0 && j == 0) return s1 - s2;
   return j == 0 ? k : j;
}

static void put_word(FILE *f, const char *s)          /* "W=%-19s " with ReWriteString(.., NULL, ESCAPE_CHAR): quotes only when needed */
{
   char buf[1100]; int n = 0, need = 0;
   for (const char *p = s; *p; p++) if (*p == ' ' || *p == '"' || *p == '\'' || *p == '\\' || (unsigned char)*p < 33 || (unsigned char)*p > 126) need = 1;
   if (!need) { fprintf(f, "W=%-19s ", s); return; }
   buf[n++] = '"';
   for (const char *p = s; *p && n < 1090; p++) { if (*p == '"' || *p == '\\') buf[n++] = '\\'; buf[n++] = *p; }
   buf[n++] = '"'; buf[n] = 0;
   fprintf(f, "W=%-19s ", buf);
}

/* LArcTotLike (HNet.h:257): the three scaled likelihoods summed in float, the word penalty added in DOUBLE (the macro's `? 0.0 :
   wdpenalty` makes the last operand a double) -- TranscriptionFromLattice accumulates that double in its backward scores and in the
   A* search (HRec.c:2176-2290); only the label's score is a float */
static double arc_total_like(const htkamd_lattice *lat, int arc)
{
   const int pron = lat->nodePron[lat->arcEnd[arc]];
   const float ac = lat->arcAc[arc] * 1.0f, lm = lat->arcLm[arc] * lat->lmScale, pr = lat->arcPr[arc] * lat->prScale;
   return (double)((ac + lm) + pr) + (pron < 0 ? 0.0 : (double)lat->wordPen);
}

float htkamd_lattice_arc_score(const htkamd_lattice *lat, int arc) { return (float)arc_total_like(lat, arc); }

int htkamd_lattice_write(const htkamd_lattice *lat, const struct htkamd_net *net, const char *path, const char *utterance, const char *lmName,
                         const char *vocabName, int format)
{
   if (format & HTKAMD_LAT_ALIGN) { htkamd_set_error("lattice_write: -q d wants the alignment records (htkamd_lattice_write_align)"); return HTKAMD_EINVAL; }
   return htkamd_lattice_write_align(lat, NULL, NULL, net, path, utterance, lmName, vocabName, format);
}

int htkamd_lattice_write_align(const htkamd_lattice *lat, const htkamd_lattice_align *al, const struct htkamd_mmf *hmms, const struct htkamd_net *net, const char *path,
                               const char *utterance, const char *lmName, const char *vocabName, int format)
{
   if (!lat || !net || !path || lat->nNodes < 2 || lat->nArcs < 1 || (al && !hmms)) { htkamd_set_error("lattice_write: bad argument"); return HTKAMD_EINVAL; }
   if (format & (HTKAMD_LAT_LBIN | HTKAMD_LAT_ALABS)) { htkamd_set_error("lattice_write: -q A / B are not supported"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "w");
   if (!f) { htkamd_set_error("lattice_write: cannot create %s", path); return HTKAMD_EIO; }
   fprintf(f, "VERSION=1.0\n");
   if (utterance) fprintf(f, "UTTERANCE=%s\n", utterance);
   if (lmName) fprintf(f, "lmname=%s\nlmscale=%-6.2f wdpenalty=%-6.2f\n", lmName, lat->lmScale, lat->wordPen);
   if (format & HTKAMD_LAT_PRLIKE) fprintf(f, "prscale=%-6.2f\n", lat->prScale);
   if (format & HTKAMD_LAT_ACLIKE) fprintf(f, "acscale=%-6.2f\n", 1.0f);
   if (vocabName) fprintf(f, "vocab=%s\n", vocabName);
   fprintf(f, "N=%-4d L=%-5d\n", lat->nNodes, lat->nArcs);
   const int nn = lat->nNodes, na = lat->nArcs;
   int *order = (int *)malloc(sizeof(int) * (size_t)(nn > na ? nn : na) + 4), *rorder = (int *)malloc(sizeof(int) * (size_t)nn);
   for (int i = 0; i < nn; i++) order[i] = i;
   g_lat = lat;
   qsort(order, (size_t)nn, sizeof(int), cmp_nodes);
   for (int i = 0; i < nn; i++) {
      const int n = order[i], pron = lat->nodePron[n];
      rorder[n] = i;
      fprintf(f, "I=%-4d ", i);
      if (format & HTKAMD_LAT_TIMES) fprintf(f, "t=%-5.2f ", (float)(lat->nodeFrame[n] * lat->frameDur));
      if (pron >= 0) {
         put_word(f, htkamd_net_word_name(net, pron));
         if (format & HTKAMD_LAT_PRON) fprintf(f, "v=%-2d ", htkamd_net_pron_num(net, pron));
      } else fprintf(f, "W=%-19s ", "!NULL");
      fprintf(f, "\n");
   }
   for (int i = 0; i < na; i++) order[i] = i;
   g_rank = rorder;
   qsort(order, (size_t)na, sizeof(int), cmp_arcs);
   for (int i = 0; i < na; i++) {
      const int a = order[i];
      fprintf(f, "J=%-5d S=%-4d E=%-4d ", i, rorder[lat->arcStart[a]], rorder[lat->arcEnd[a]]);
      if (format & HTKAMD_LAT_ACLIKE) fprintf(f, "a=%-9.2f ", lat->arcAc[a]);
      if (format & HTKAMD_LAT_LMLIKE) fprintf(f, "l=%-7.3f ", lat->arcLm[a]);
      if (format & HTKAMD_LAT_PRLIKE) fprintf(f, "r=%-6.2f ", lat->arcPr[a]);
      if (al && (format & HTKAMD_LAT_ALIGN) && al->arcAlignOff[a + 1] > al->arcAlignOff[a]) {      /* OutputAlign (HNet.c:503-516) */
         fprintf(f, "d=:");
         for (int q = al->arcAlignOff[a]; q < al->arcAlignOff[a + 1]; q++) {
            const char *mn = htkamd_mmf_phys_name(hmms, al->alModel[q]);
            if (al->alState[q] < 0) fprintf(f, "%s", mn ? mn : "?");
            else if (al->models) fprintf(f, "s%d", al->alState[q]);
            else fprintf(f, "%s[%d]", mn ? mn : "?", al->alState[q]);
            if (format & HTKAMD_LAT_ALDUR) fprintf(f, ",%.2f", (float)(al->alDur[q] * lat->frameDur));
            if (format & HTKAMD_LAT_ALLIKE) fprintf(f, ",%.2f", al->alLike[q]);
            fprintf(f, ":");
         }
      }
      fprintf(f, "\n");
   }
   free(order); free(rorder);
   g_lat = NULL; g_rank = NULL;
   if (fclose(f)) { htkamd_set_error("lattice_write: write error on %s", path); return HTKAMD_EIO; }
   return HTKAMD_OK;
}

/* TranscriptionFromLattice's label list for ONE alternative whose arcs carry alignment records (HRec.c:2284-2338): a label per model
   record (-m) or per state record (-f, the model then an auxiliary label of its first state), the WORD as the last auxiliary label of the
   arc's first label, scores = the records' likelihoods, times accumulated from the arc's start in the records' float durations. */
int htkamd_lattice_align_trans(const htkamd_lattice *lat, const htkamd_lattice_align *al, const struct htkamd_mmf *hmms, const struct htkamd_net *net,
                               const int *arcs, int nArcs, htkamd_trans **out)
{
   if (!lat || !al || !hmms || !net || !arcs || nArcs < 0 || !out) { htkamd_set_error("lattice_align_trans: bad argument"); return HTKAMD_EINVAL; }
   int states = 0, models = 0;
   for (int i = 0; i < nArcs; i++) {
      const int a = arcs[i];
      if (lat->nodePron[lat->arcEnd[a]] < 0) continue;                      /* !NULL */
      if (al->arcAlignOff[a + 1] == al->arcAlignOff[a]) { *out = NULL; return HTKAMD_OK; }      /* an arc without records: word labels (the caller's) */
      for (int q = al->arcAlignOff[a]; q < al->arcAlignOff[a + 1]; q++) { if (al->alState[q] < 0) models = 1; else if (al->alState[q] > 0) states = 1; }
   }
   const int nAux = states + models;
   if (nAux == 0) { *out = NULL; return HTKAMD_OK; }
   htkamd_trans *tr;
   int rc = htkamd_trans_create(nAux, &tr);
   if (rc) return rc;
   for (int i = 0; i < nArcs; i++) {
      const int a = arcs[i], pron = lat->nodePron[lat->arcEnd[a]];
      if (pron < 0) continue;
      const char *word = htkamd_net_word_name(net, pron), *model = NULL;
      const float lmf = lat->arcLm[a] * lat->lmScale;                      /* LArcTotLMLike (HNet.h:252) */
      float lm = (float)((double)lmf + (double)lat->wordPen), modlk = 0.0f;
      double start = (lat->nodeFrame[lat->arcStart[a]] * lat->frameDur) * 1.0E7;
      for (int q = al->arcAlignOff[a]; q < al->arcAlignOff[a + 1]; q++) {
         const char *mn = htkamd_mmf_phys_name(hmms, al->alModel[q]);
         char buf[300];
         const char *label;
         if (al->alState[q] < 0) label = mn;
         else { if (al->models) snprintf(buf, sizeof(buf), "s%d", al->alState[q]); else snprintf(buf, sizeof(buf), "%s[%d]", mn ? mn : "?", al->alState[q]); label = buf; }
         if (al->alState[q] < 0 && states) { model = label; modlk = al->alLike[q]; continue; }
         const float dur = (float)(al->alDur[q] * lat->frameDur);
         const double end = start + dur * 1.0E7;
         const char *a1 = NULL, *a2 = NULL; float s1 = 0.0f, s2 = 0.0f;
         if (models && states) { a1 = model; s1 = modlk; model = NULL; modlk = 0.0f; a2 = word; s2 = lm; }
         else { a1 = word; s1 = lm; }
         rc = htkamd_trans_add(tr, start, end, label, al->alLike[q], a1, s1, a2, s2);
         if (rc) { htkamd_trans_free(tr); return rc; }
         start = end; word = NULL; lm = 0.0f;
      }
   }
   *out = tr;
   return HTKAMD_OK;
}

/* ------------------------------------------------------------------------------------------ N best through the lattice */
typedef struct nbe { int link, knil, prev; double score, like; int lnode, larc; } nbe;

static int word_match(const nbe *e, int cmp, int ans, const htkamd_lattice *lat, const struct htkamd_net *net)
{
   for (;;) {
      if (cmp == ans) return 1;
      if (cmp < 0 || ans < 0) return 0;
      const int pc = lat->nodePron[lat->arcEnd[e[cmp].larc]], pa = lat->nodePron[lat->arcEnd[e[ans].larc]];
      if ((pc < 0) != (pa < 0)) return 0;
      if (pc >= 0 && strcmp(htkamd_net_word_name(net, pc), htkamd_net_word_name(net, pa))) return 0;
      cmp = e[cmp].prev; ans = e[ans].prev;
   }
}

int htkamd_lattice_nbest(const htkamd_lattice *lat, const struct htkamd_net *net, int N, int maxLen, int *nAlt, int *altLen, int *altArcs)
{
   if (!lat || !net || N < 1 || maxLen < 1 || !nAlt || !altLen || !altArcs || lat->nNodes < 2) { htkamd_set_error("lattice_nbest: bad argument"); return HTKAMD_EINVAL; }
   const int nn = lat->nNodes, na = lat->nArcs;
   /* adjacency: arcs leaving / entering a node, most recently added first (the reference prepends: la->farc = ns->foll) */
   int *foll = (int *)malloc(sizeof(int) * (size_t)nn), *farc = (int *)malloc(sizeof(int) * (size_t)na), *pred = (int *)malloc(sizeof(int) * (size_t)nn), *parc = (int *)malloc(sizeof(int) * (size_t)na);
   for (int i = 0; i < nn; i++) foll[i] = pred[i] = -1;
   for (int a = 0; a < na; a++) { farc[a] = foll[lat->arcStart[a]]; foll[lat->arcStart[a]] = a; parc[a] = pred[lat->arcEnd[a]]; pred[lat->arcEnd[a]] = a; }
   double *score = (double *)malloc(sizeof(double) * (size_t)nn);
   int *num = (int *)malloc(sizeof(int) * (size_t)nn), *order = (int *)malloc(sizeof(int) * (size_t)nn), n = 0;
   for (int i = 0; i < nn; i++) { score[i] = (foll[i] < 0) ? 0.0 : LZERO; num[i] = -1; }
   {  /* MarkBack (HRec.c:2157): post-order over predecessors, iteratively */
      int *st = (int *)malloc(sizeof(int) * (size_t)(2 * nn + 2)), sp;
      for (int i = 0; i < nn; i++) {
         if (num[i] != -1) continue;
         sp = 0; st[sp++] = i; st[sp++] = pred[i]; num[i] = -2;
         while (sp > 0) {
            const int a = st[sp - 1], node = st[sp - 2];
            if (a < 0) { num[node] = n++; sp -= 2; continue; }
            st[sp - 1] = parc[a];
            const int s = lat->arcStart[a];
            if (num[s] == -1) { num[s] = -2; st[sp++] = s; st[sp++] = pred[s]; }
         }
      }
      free(st);
   }
   for (int i = 0; i < nn; i++) order[num[i]] = i;
   for (int i = nn - 1; i > 0; i--) {
      const int ln = order[i];
      for (int a = pred[ln]; a >= 0; a = parc[a]) {
         const double sc = score[ln] + arc_total_like(lat, a);
         if (sc > score[lat->arcStart[a]]) score[lat->arcStart[a]] = sc;
      }
   }
   int cap = 1024, ne = 2;                               /* entries 0 = head, 1 = tail */
   nbe *e = (nbe *)malloc(sizeof(nbe) * (size_t)cap);
   e[0].link = 1; e[0].knil = -1; e[1].link = -1; e[1].knil = 0; e[0].score = e[1].score = LZERO;
#define PUSH(sc_, lk_, node_, arc_, prev_) do { if (ne + 1 > cap) { cap *= 2; e = (nbe *)realloc(e, sizeof(nbe) * (size_t)cap); } \
      nbe *x_ = &e[ne]; x_->score = (sc_); x_->like = (lk_); x_->lnode = (node_); x_->larc = (arc_); x_->prev = (prev_); \
      int pos_ = e[0].link; while ((sc_) < e[pos_].score) pos_ = e[pos_].link; \
      x_->knil = e[pos_].knil; x_->link = pos_; e[x_->knil].link = ne; e[pos_].knil = ne; ne++; } while (0)
   int rc = HTKAMD_OK;
   for (int i = 0; i < nn && !rc; i++) {
      if (pred[i] >= 0) continue;
      if (score[i] < LSMALL) { htkamd_set_error("lattice_nbest: no route through the lattice"); rc = HTKAMD_EMODEL; break; }
      for (int a = foll[i]; a >= 0; a = farc[a]) {
         const double like = arc_total_like(lat, a), sc = like + score[lat->arcEnd[a]];
         if (sc < LSMALL) continue;
         PUSH(sc, like, lat->arcEnd[a], a, -1);
      }
   }
   int *ans = (int *)malloc(sizeof(int) * (size_t)N), nAns = 0;
   while (!rc && nAns < N && e[0].link != 1) {
      const int best = e[0].link;
      e[e[best].link].knil = e[best].knil; e[e[best].knil].link = e[best].link;
      if (foll[e[best].lnode] >= 0) {
         for (int a = foll[e[best].lnode]; a >= 0; a = farc[a]) {
            const double like = e[best].like + arc_total_like(lat, a), sc = like + score[lat->arcEnd[a]];
            if (sc < LSMALL) continue;
            PUSH(sc, like, lat->arcEnd[a], a, best);
         }
         continue;
      }
      int dup = 0;
      for (int i = 0; i < nAns && !dup; i++) dup = word_match(e, best, ans[i], lat, net);
      if (!dup) ans[nAns++] = best;
   }
#undef PUSH
   for (int i = 0; i < nAns && !rc; i++) {
      int len = 0;
      for (int p = e[ans[i]].prev; p >= 0; p = e[p].prev) len++;       /* the arc into the end node is left out, as the reference's loops do */
      if (len > maxLen) { htkamd_set_error("lattice_nbest: an alternative has %d arcs (maxLen %d)", len, maxLen); rc = HTKAMD_EINVAL; break; }
      altLen[i] = len;
      int k = len;
      for (int p = e[ans[i]].prev; p >= 0; p = e[p].prev) altArcs[(size_t)i * maxLen + --k] = e[p].larc;
   }
   *nAlt = nAns;
   free(foll); free(farc); free(pred); free(parc); free(score); free(num); free(order); free(e); free(ans);
   return rc;
}
