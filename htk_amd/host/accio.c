/* accio.c -- host-side (C) reading and writing of HERest accumulator files (HERN.acc), so that the
 * reference's parallel mode and this library can exchange statistics:
 *     HERest -p N  ->  DumpAccs (HTrain.c:1453-1505) + float totalPr, int totalT (HERest.c:546-548)
 *     HERest -p 0  ->  LoadAccs (HTrain.c:1625-1687): ADDS each file to the accumulators
 * File layout (binary, big-endian, HShell.c:1638): per physical HMM in HMM-scan order -- quoted name + newline,
 * int32 example count, for every not-yet-seen state WtAcc {c[M], occ} followed by, per not-yet-seen Gaussian,
 * MuAcc {mu[D], occ} and VaAcc {var[D], occ}; for a not-yet-seen transition matrix TrAcc {tran[N][N], occ[N]};
 * int32 marker 123456.  A mean or variance vector shared by several Gaussians (~u / ~v macros) carries ONE MuAcc / VaAcc in the
 * reference and is dumped where the scan first meets it (IsSeenV on the vector, HTrain.c:1484-1493): the writer puts the sum over the
 * sharers there, the reader adds the file's record to the vector's first sharer (htkamd_model_update pools the sharers anyway).
 * The scan order is the order of the 'h' macros in the set's hash table (HUtil.c:265-295
 * GoNextHMM over mtab; HModel.c:3314 Hash, :3384 head insertion): ascending hash bucket, later definitions first.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

#define MACHASHSIZE 250007                      /* HModel.h:50 */

static unsigned mac_hash(const char *name)      /* HModel.c:3314-3321 */
{
   unsigned hashval;
   for (hashval = 0; *name != '\0'; name++) hashval = *name + 31 * hashval;
   return hashval % MACHASHSIZE;
}

typedef struct { unsigned h; int idx; } scan_ent;
static int scan_cmp(const void *a, const void *b)
{
   const scan_ent *x = (const scan_ent *)a, *y = (const scan_ent *)b;
   if (x->h != y->h) return (x->h < y->h) ? -1 : 1;
   return (x->idx > y->idx) ? -1 : (x->idx < y->idx);       /* head insertion: later definitions come first */
}

/* order[k] = index of the k-th physical HMM in the reference's scan (names in definition order) */
int htkamd_hmm_scan_order(const char *const *names, int H, int *order)
{
   scan_ent *e;
   int i;
   if (!names || !order || H < 0) { htkamd_set_error("hmm_scan_order: bad argument"); return HTKAMD_EINVAL; }
   e = (scan_ent *)malloc(sizeof(scan_ent) * (size_t)(H ? H : 1));
   for (i = 0; i < H; i++) { e[i].h = mac_hash(names[i]); e[i].idx = i; }
   qsort(e, (size_t)H, sizeof(scan_ent), scan_cmp);
   for (i = 0; i < H; i++) order[i] = e[i].idx;
   free(e);
   return HTKAMD_OK;
}

int htkamd_accs_layout_from_desc(const htkamd_model_desc *d, htkamd_accs_layout *lay)
{
   size_t o = 0, GD;
   int t, sumN = 0;
   if (!d || !lay) { htkamd_set_error("accs_layout_from_desc: NULL"); return HTKAMD_EINVAL; }
   GD = (size_t)d->numGauss * d->vecSize;
   for (t = 0; t < d->numTrans; t++) sumN += d->transN[t];
   lay->mu = o; o += GD;
   lay->muOcc = o; o += d->numGauss;
   lay->va = o; o += GD;
   lay->vaOcc = o; o += d->numGauss;
   lay->wt = o; o += d->numComp;
   lay->wtOcc = o; o += (size_t)d->numStates * (d->numStreams > 1 ? d->numStreams : 1);      /* one WtAcc per (state, stream) */
   lay->tr = o; o += d->transOff[d->numTrans];
   lay->trOcc = o; o += sumN;
   lay->nEgs = o; o += d->numPhys;
   lay->totalPr = o++; lay->totalT = o++; lay->nUttDone = o++; lay->nUttSkipped = o++; lay->nEval = o++;
   lay->total = o;
   return HTKAMD_OK;
}

static void put_be32(FILE *f, const void *p)
{
   const unsigned char *b = (const unsigned char *)p;
   unsigned char o[4] = {b[3], b[2], b[1], b[0]};
   fwrite(o, 1, 4, f);
}
static void put_f(FILE *f, double v) { float x = (float)v; put_be32(f, &x); }
static void put_i(FILE *f, int v) { put_be32(f, &v); }
static int get_be32(FILE *f, void *p)
{
   unsigned char b[4], *o = (unsigned char *)p;
   if (fread(b, 1, 4, f) != 4) return 0;
   o[0] = b[3]; o[1] = b[2]; o[2] = b[1]; o[3] = b[0];
   return 1;
}

/* walks the file structure once; `wr` != 0 writes vec -> file, else adds file -> vec */
static int acc_walk(FILE *f, int wr, const htkamd_model_desc *d, const htkamd_accs_layout *lay, double *vec,
                    const char *const *names, int uFlags, const char *path, const int *meanShare, const int *varShare)
{
   const int D = d->vecSize, NS = d->numStreams > 1 ? d->numStreams : 1;
   unsigned char *seenMu = (unsigned char *)calloc((size_t)d->numGauss + 1, 1), *seenVa = (unsigned char *)calloc((size_t)d->numGauss + 1, 1);
   double *pool = NULL;                         /* writer, shared vectors: statistics summed into the first sharer */
   unsigned char *seenS = (unsigned char *)calloc((size_t)d->numStates, 1);
   unsigned char *seenG = (unsigned char *)calloc((size_t)d->numGauss, 1);
   unsigned char *seenT = (unsigned char *)calloc((size_t)d->numTrans, 1);
   int *order = (int *)malloc(sizeof(int) * (size_t)(d->numPhys ? d->numPhys : 1));
   int *occOff = (int *)calloc((size_t)d->numTrans + 1, sizeof(int));
   int k, j, c, i, rc = HTKAMD_OK, t;
   float x;
   for (t = 0; t < d->numTrans; t++) occOff[t + 1] = occOff[t] + d->transN[t];
   htkamd_hmm_scan_order(names, d->numPhys, order);
   /* vector ids -> the first Gaussian that uses the vector */
   int *mLead = NULL, *vLead = NULL;
   for (t = 0; t < 2; t++) {
      const int *share = t ? varShare : meanShare;
      int g, mx = -1, *lead, *first;
      if (!share) continue;
      lead = (int *)malloc(sizeof(int) * (size_t)(d->numGauss + 1));
      for (g = 0; g < d->numGauss; g++) if (share[g] > mx) mx = share[g];
      first = (int *)malloc(sizeof(int) * (size_t)(mx + 2));
      for (g = 0; g <= mx; g++) first[g] = -1;
      for (g = 0; g < d->numGauss; g++) {
         lead[g] = g;
         if (share[g] >= 0) { if (first[share[g]] < 0) first[share[g]] = g; lead[g] = first[share[g]]; }
      }
      free(first);
      if (t) vLead = lead; else mLead = lead;
   }
#define MLD(g) (mLead ? mLead[g] : (g))
#define VLD(g) (vLead ? vLead[g] : (g))
   if (wr && (mLead || vLead)) {
      int g;
      pool = (double *)malloc(sizeof(double) * lay->total);
      memcpy(pool, vec, sizeof(double) * lay->total);
      for (g = 0; g < d->numGauss; g++) {
         if (MLD(g) != g) { for (i = 0; i < D; i++) pool[lay->mu + (size_t)MLD(g) * D + i] += vec[lay->mu + (size_t)g * D + i]; pool[lay->muOcc + MLD(g)] += vec[lay->muOcc + g]; }
         if (VLD(g) != g) { for (i = 0; i < D; i++) pool[lay->va + (size_t)VLD(g) * D + i] += vec[lay->va + (size_t)g * D + i]; pool[lay->vaOcc + VLD(g)] += vec[lay->vaOcc + g]; }
      }
      vec = pool;
   }
#define IO(off)  do { if (wr) put_f(f, vec[off]); else { if (!get_be32(f, &x)) { rc = HTKAMD_EINVAL; goto bad; } vec[off] += (double)x; } } while (0)
   for (k = 0; k < d->numPhys; k++) {
      const int h = order[k], ti = d->hmmTrans[h], N = d->transN[ti];
      if (wr) {
         const char *p;
         fputc('"', f);
         for (p = names[h]; *p; p++) { if (*p == '"' || *p == '\\') fputc('\\', f); fputc(*p, f); }
         fputc('"', f); fputc('\n', f);
         put_i(f, (int)(vec[lay->nEgs + h] + 0.5));
      } else {
         char buf[512]; int n = 0, ch, negs;
         if (fgetc(f) != '"') { rc = HTKAMD_EINVAL; goto bad; }
         while ((ch = fgetc(f)) != EOF && ch != '"' && n < 510) { if (ch == '\\') ch = fgetc(f); buf[n++] = (char)ch; }
         buf[n] = 0;
         if (ch != '"' || fgetc(f) != '\n' || strcmp(buf, names[h]) != 0) {      /* CheckPName HTrain.c:1600 */
            htkamd_set_error("accs_load_file: %s: expected HMM \"%s\", found \"%s\"", path, names[h], buf);
            rc = HTKAMD_EINVAL; goto bad2;
         }
         if (!get_be32(f, &negs)) { rc = HTKAMD_EINVAL; goto bad; }
         vec[lay->nEgs + h] += negs;
      }
      for (j = 0; j < N - 2; j++) {
         const int s = d->hmmState[d->hmmStateOff[h] + j];
         int ks;
         if (seenS[s]) continue;
         seenS[s] = 1;
         for (ks = 0; ks < NS; ks++) {           /* GoNextStream inside GoNextState: the stream's WtAcc, then its Gaussians (vectors of the stream's width) */
            const int e = s * NS + ks;
            for (c = d->stateCompOff[e]; c < d->stateCompOff[e + 1]; c++) IO(lay->wt + c);
            IO(lay->wtOcc + e);
            if (d->hsKind == HTKAMD_HS_TIED) continue;   /* tied mixtures: the pool's MuAcc / VaAcc follow the last model (below) */
            for (c = d->stateCompOff[e]; c < d->stateCompOff[e + 1]; c++) {
               const int g = d->compGauss[c], gm = MLD(g), gv = VLD(g);
               if (seenG[g]) continue;              /* a shared mixture pdf (~m) */
               seenG[g] = 1;
#define INS(i) (NS == 1 || d->dimStream[i] == ks)
               if ((uFlags & HTKAMD_UPMEANS) && !seenMu[gm]) { seenMu[gm] = 1; for (i = 0; i < D; i++) if (INS(i)) IO(lay->mu + (size_t)gm * D + i); IO(lay->muOcc + gm); }
               if ((uFlags & HTKAMD_UPVARS) && !seenVa[gv]) { seenVa[gv] = 1; for (i = 0; i < D; i++) if (INS(i)) IO(lay->va + (size_t)gv * D + i); IO(lay->vaOcc + gv); }
#undef INS
            }
         }
      }
      if (!seenT[ti]) {
         seenT[ti] = 1;
         for (i = 0; i < N * N; i++) IO(lay->tr + d->transOff[ti] + i);
         for (i = 0; i < N; i++) IO(lay->trOcc + occOff[ti] + i);
      }
      if (wr) put_i(f, 123456);
      else { int mark; if (!get_be32(f, &mark) || mark != 123456) { htkamd_set_error("accs_load_file: %s: marker missing after \"%s\"", path, names[h]); rc = HTKAMD_EINVAL; goto bad2; } }
   }
   if (d->hsKind == HTKAMD_HS_TIED) {           /* DumpAccs / LoadAccs, TIEDHS tail (HTrain.c:1493-1501,1675-1684): MuAcc then VaAcc of every pool
                                                   Gaussian, stream by stream, whatever the update flags say */
      int ks;
      for (ks = 0; ks < NS; ks++)
         for (c = d->stateCompOff[ks]; c < d->stateCompOff[ks + 1]; c++) {      /* the pool as state 0 lists it */
            const int g = d->compGauss[c];
            for (i = 0; i < D; i++) if (NS == 1 || d->dimStream[i] == ks) IO(lay->mu + (size_t)g * D + i);
            IO(lay->muOcc + g);
            for (i = 0; i < D; i++) if (NS == 1 || d->dimStream[i] == ks) IO(lay->va + (size_t)g * D + i);
            IO(lay->vaOcc + g);
         }
   }
   if (wr) { put_f(f, vec[lay->totalPr]); put_i(f, (int)(vec[lay->totalT] + 0.5)); }
   else { int tt; if (!get_be32(f, &x) || !get_be32(f, &tt)) { rc = HTKAMD_EINVAL; goto bad; } vec[lay->totalPr] += (double)x; vec[lay->totalT] += tt; }
   goto done;
bad:
   htkamd_set_error("accs_load_file: %s: truncated or malformed", path);
bad2:
done:
#undef IO
#undef MLD
#undef VLD
   free(pool); free(seenMu); free(seenVa); free(mLead); free(vLead);
   free(seenS); free(seenG); free(seenT); free(order); free(occOff);
   return rc;
}

/* DumpAccs + HERest's trailer: write the host vector as HER<n>.acc */
int htkamd_accs_dump_file(const htkamd_model_desc *d, const double *vec, const char *const *names, int uFlags, const char *path)
{
   return htkamd_accs_dump_file_shared(d, vec, names, uFlags, NULL, NULL, path);
}
int htkamd_accs_dump_file_shared(const htkamd_model_desc *d, const double *vec, const char *const *names, int uFlags, const int *meanShare, const int *varShare, const char *path)
{
   htkamd_accs_layout lay;
   FILE *f;
   int rc;
   if (!d || !vec || !names || !path) { htkamd_set_error("accs_dump_file: NULL argument"); return HTKAMD_EINVAL; }
   htkamd_accs_layout_from_desc(d, &lay);
   f = fopen(path, "wb");
   if (!f) { htkamd_set_error("accs_dump_file: cannot open %s", path); return HTKAMD_EINVAL; }
   rc = acc_walk(f, 1, d, &lay, (double *)vec, names, uFlags, path, meanShare, varShare);
   fclose(f);
   return rc;
}

/* LoadAccs + trailer: ADD the file to the host vector */
int htkamd_accs_load_file(const htkamd_model_desc *d, double *vec, const char *const *names, int uFlags, const char *path)
{
   return htkamd_accs_load_file_shared(d, vec, names, uFlags, NULL, NULL, path);
}
int htkamd_accs_load_file_shared(const htkamd_model_desc *d, double *vec, const char *const *names, int uFlags, const int *meanShare, const int *varShare, const char *path)
{
   htkamd_accs_layout lay;
   FILE *f;
   int rc;
   if (!d || !vec || !names || !path) { htkamd_set_error("accs_load_file: NULL argument"); return HTKAMD_EINVAL; }
   htkamd_accs_layout_from_desc(d, &lay);
   f = fopen(path, "rb");
   if (!f) { htkamd_set_error("accs_load_file: cannot open %s", path); return HTKAMD_EINVAL; }
   rc = acc_walk(f, 0, d, &lay, vec, names, uFlags, path, meanShare, varShare);
   if (rc == HTKAMD_OK && fgetc(f) != EOF) { htkamd_set_error("accs_load_file: %s: trailing bytes", path); rc = HTKAMD_EINVAL; }
   fclose(f);
   return rc;
}

/* HERest -s: the state occupation statistics file HHEd's clustering commands read (StatReport / PrintStats, HERest.c:708-747):
   one line per physical HMM in HMM-scan order, "%4d %14s %4d " index, quoted name (ReWriteString with double quotes), number of
   training examples, then " %10f" with the occupation count of every emitting state (the state's WtAcc occ, a float). */
int htkamd_stats_write_file(const htkamd_model_desc *d, const double *vec, const char *const *names, const char *path)
{
   if (!d || !vec || !names || !path) { htkamd_set_error("stats_write_file: NULL argument"); return HTKAMD_EINVAL; }
   htkamd_accs_layout lay;
   int rc = htkamd_accs_layout_from_desc(d, &lay);
   if (rc) return rc;
   FILE *f = fopen(path, "w");
   if (!f) { htkamd_set_error("stats_write_file: cannot create %s", path); return HTKAMD_EIO; }
   int *order = (int *)malloc(sizeof(int) * (size_t)(d->numPhys ? d->numPhys : 1));
   htkamd_hmm_scan_order(names, d->numPhys, order);
   for (int k = 0; k < d->numPhys; k++) {
      const int h = order[k], N = d->transN[d->hmmTrans[h]];
      char buf[1024]; int n = 0;
      buf[n++] = '"';
      for (const char *p = names[h]; *p && n < 1000; p++) { if (*p == '"' || *p == '\\') buf[n++] = '\\'; buf[n++] = *p; }
      buf[n++] = '"'; buf[n] = 0;
      fprintf(f, "%4d %14s %4d ", k + 1, buf, (int)(vec[lay.nEgs + h] + 0.5));
      const int NS = d->numStreams > 1 ? d->numStreams : 1;
      for (int j = 0; j < N - 2; j++) fprintf(f, " %10f", (float)vec[lay.wtOcc + (size_t)d->hmmState[d->hmmStateOff[h] + j] * NS]);      /* the first stream's WtAcc (PrintStats HERest.c:690) */
      fprintf(f, "\n");
   }
   free(order);
   fclose(f);
   return HTKAMD_OK;
}
