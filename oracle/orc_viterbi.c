/* orc_viterbi.c -- CPU restatement of HRec's 1-best token passing for a forced alignment (TEST INFRASTRUCTURE).
 *
 * HVite -a builds, from a label file, a linear network: the HMM nodes of word 1, its word-end node, the HMM nodes
 * of word 2, ... (LatticeFromLabels HNet.c:1516 + ExpandWordNet HNet.c:3438; one model per word in the synthetic
 * sets, one or more in general -- the chain of PHYSICAL MODELS is what this file takes).  On such a chain the
 * machinery of HRec.c reduces to:
 *   pass 1  StepHMM1 (HRec.c:642-787): for every emitting state j the best predecessor over i in
 *           [seIndex[j][0], seIndex[j][1]] (CreateSEIndex HRec.c:1403), FIRST maximum wins (strict >), tokens are
 *           double sums of float terms; if best > genThresh add the output probability (cSOutP arithmetic),
 *           else the token dies; the entry token is consumed; exit = best of like_i + a_iN over seIndex[N].
 *   thresholds (HRec.c:1997-2004): genThresh = float(genMax - genBeam), floored at LSMALL, used by pass 2 of this
 *           frame and pass 1 of the NEXT frame.
 *   pass 2  (HRec.c:2007-2016) in chain order: an instance whose max is below genThresh is detached (all its tokens
 *           die); otherwise tee models pass entry -> exit (StepHMM2 HRec.c:790), and the exit token, if above
 *           genThresh, becomes the entry token of the next model (SetEntryState HRec.c:1303: strict >).
 *           Word-end nodes add wordpen + pronprob*pscale = 0 here and lm = 0.
 * Traceback = the Align records (state entry: like before the output prob, frame-1; model end: exit like, frame;
 * HRec.c:691-715,767-771) turned into label segments by LatFromPaths (HRec.c:1512-1660): segment score =
 * like(next record) - like(this record).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "htk_oracle.h"

typedef struct {
   int N, slot0;
   const float *tp;
   int seLo[16], seHi[16];      /* per destination state j (2..N): predecessor range */
   int tee;
} vmodel;

#define TPM(m,i,j) ((m)->tp[((i)-1)*(m)->N + ((j)-1)])

/* returns number of segments written (one per visited emitting state), or -1 if no token survived.
   seg*: [maxSeg] arrays; model-level results: modStart/modEnd/modScore [Q] (modStart = -1 for skipped tee models) */
int orc_viterbi_align(const orc_model *m, const float *X, int T, const int *labs, int Q, float genBeam,
                      int maxSeg, int *segQ, int *segState, int *segStart, int *segEnd, double *segScore,
                      int *modStart, int *modEnd, double *modScore, double *totalLike)
{
   vmodel *vm = (vmodel *)calloc((size_t)Q + 2, sizeof(vmodel));
   int q, i, j, t, nSlots = 0, maxN = 0, nSeg = -1;
   double *like, *nw, *exitL, *entL, *pre, *exl, *entAt;
   signed char *bp, *exbp;
   unsigned char *active;
   double *instMax;
   float genThresh = (float)ORC_LSMALL;

   for (q = 1; q <= Q; q++) {
      int ti = m->hmmTrans[labs[q - 1]];
      vmodel *v = vm + q;
      v->N = m->transN[ti]; v->tp = m->transP + m->transOff[ti]; v->slot0 = nSlots;
      nSlots += v->N - 2;
      if (v->N > maxN) maxN = v->N;
      if (v->N > 15) { free(vm); return -2; }
      for (j = 2; j <= v->N; j++) {                        /* CreateSEIndex */
         int mn, mx;
         for (mn = (j == v->N) ? 2 : 1; mn < v->N; mn++) if (TPM(v, mn, j) > ORC_LSMALL) break;
         for (mx = v->N - 1; mx > 1; mx--) if (TPM(v, mx, j) > ORC_LSMALL) break;
         if (mn > mx) { mn = (j == v->N) ? 2 : 1; mx = v->N - 1; }
         v->seLo[j] = mn; v->seHi[j] = mx;
      }
      v->tee = TPM(v, 1, v->N) > ORC_LSMALL;
   }
   like = (double *)malloc(sizeof(double) * (size_t)(Q + 2) * (maxN + 1));
   nw = (double *)malloc(sizeof(double) * (size_t)(maxN + 1));
   exitL = (double *)malloc(sizeof(double) * (size_t)(Q + 2));
   entL = (double *)malloc(sizeof(double) * (size_t)(Q + 2));
   instMax = (double *)malloc(sizeof(double) * (size_t)(Q + 2));
   active = (unsigned char *)calloc((size_t)Q + 2, 1);
   pre = (double *)malloc(sizeof(double) * (size_t)(T + 1) * (nSlots ? nSlots : 1));
   bp = (signed char *)malloc((size_t)(T + 1) * (nSlots ? nSlots : 1));
   exl = (double *)malloc(sizeof(double) * (size_t)(T + 1) * (Q + 2));
   entAt = (double *)malloc(sizeof(double) * (size_t)(T + 1) * (Q + 2));
   exbp = (signed char *)malloc((size_t)(T + 1) * (Q + 2));
#define LK(q,i) like[(size_t)(q) * (maxN + 1) + (i)]
   for (q = 0; q <= Q + 1; q++) { for (i = 0; i <= maxN; i++) LK(q, i) = ORC_LZERO; exitL[q] = ORC_LZERO; entL[q] = ORC_LZERO; instMax[q] = ORC_LZERO; }

   /* StartRecognition (HRec.c:1884-1932): the initial node's token (like 0) is propagated in a pass 2 at frame 0 */
   for (t = 0; t <= T; t++) {
      double genMax = ORC_LZERO;
      if (t >= 1) {
         /* ---- pass 1 ---- */
         for (q = 1; q <= Q; q++) {
            vmodel *v = vm + q;
            double mx = ORC_LZERO;
            if (!active[q]) { exl[(size_t)t * (Q + 2) + q] = ORC_LZERO; exbp[(size_t)t * (Q + 2) + q] = 0; continue; }
            for (j = 2; j < v->N; j++) {
               int arg = v->seLo[j];
               double best = LK(q, arg) + TPM(v, arg, j);
               for (i = arg + 1; i <= v->seHi[j]; i++) {
                  double c = LK(q, i) + TPM(v, i, j);
                  if (c > best) { best = c; arg = i; }
               }
               pre[(size_t)t * nSlots + v->slot0 + j - 2] = best;
               bp[(size_t)t * nSlots + v->slot0 + j - 2] = (signed char)arg;
               if (best > genThresh) {
                  int s = m->hmmState[m->hmmStateOff[labs[q - 1]] + (j - 2)];
                  float outp = orc_state_outp(m, s, X + (size_t)(t - 1) * m->D, NULL);
                  nw[j] = best + outp;
                  if (nw[j] > mx) mx = nw[j];
               } else nw[j] = ORC_LZERO;
            }
            LK(q, 1) = ORC_LZERO;                           /* entry token consumed */
            for (j = 2; j < v->N; j++) LK(q, j) = nw[j];
            instMax[q] = mx;
            if (mx > genMax) genMax = mx;
            {
               int arg = v->seLo[v->N];
               double best = LK(q, arg) + TPM(v, arg, v->N);
               for (i = arg + 1; i <= v->seHi[v->N]; i++) {
                  double c = LK(q, i) + TPM(v, i, v->N);
                  if (c > best) { best = c; arg = i; }
               }
               if (best > ORC_LSMALL) { exitL[q] = best; exbp[(size_t)t * (Q + 2) + q] = (signed char)arg; }
               else { exitL[q] = ORC_LZERO; exbp[(size_t)t * (Q + 2) + q] = 0; }
            }
         }
         {  /* thresholds */
            genThresh = (float)(genMax - genBeam);
            if (genThresh < ORC_LSMALL) genThresh = (float)ORC_LSMALL;
         }
      }
      /* ---- pass 2 (at t == 0: only the initial token enters model 1) ---- */
      {
         double carry = (t == 0) ? 0.0 : ORC_LZERO;         /* token offered to the next model's entry state */
         int haveCarry = (t == 0);
         for (q = 1; q <= Q; q++) {
            vmodel *v = vm + q;
            if (haveCarry && carry > genThresh) {           /* SetEntryState */
               if (!active[q]) { active[q] = 1; instMax[q] = ORC_LZERO; for (i = 1; i < v->N; i++) LK(q, i) = ORC_LZERO; exitL[q] = ORC_LZERO; }
               if (carry > LK(q, 1)) LK(q, 1) = carry;
               if (LK(q, 1) > instMax[q]) instMax[q] = LK(q, 1);
            }
            entAt[(size_t)t * (Q + 2) + q] = active[q] ? LK(q, 1) : ORC_LZERO;
            haveCarry = 0; carry = ORC_LZERO;
            if (!active[q]) continue;
            if (t >= 1 || LK(q, 1) > ORC_LSMALL) {
               if (instMax[q] < genThresh) {                /* DetachInst */
                  active[q] = 0;
                  for (i = 1; i < v->N; i++) LK(q, i) = ORC_LZERO;
                  exitL[q] = ORC_LZERO;
                  exl[(size_t)t * (Q + 2) + q] = ORC_LZERO;
                  continue;
               }
               if (v->tee) {                                 /* StepHMM2 */
                  double c = LK(q, 1) + TPM(v, 1, v->N);
                  if (c > exitL[q]) { exitL[q] = c; exbp[(size_t)t * (Q + 2) + q] = 1; }
               }
               exl[(size_t)t * (Q + 2) + q] = exitL[q];
               if (exitL[q] > genThresh) { carry = exitL[q]; haveCarry = 1; }
               if (t == 0) exitL[q] = ORC_LZERO;
            }
         }
         if (t == T) {
            /* CompleteRecognition: the token that reached the final node */
            if (haveCarry && carry > ORC_LSMALL) *totalLike = carry; else *totalLike = ORC_LZERO;
         }
      }
      /* exit tokens are rebuilt by pass 1 of the next frame */
   }

   if (*totalLike > ORC_LSMALL) {
      /* ---- traceback ---- */
      int tcur = T;
      nSeg = 0;
      for (q = 1; q <= Q; q++) { modStart[q - 1] = -1; modEnd[q - 1] = -1; modScore[q - 1] = 0.0; }
      q = Q;
      while (q >= 1) {
         vmodel *v = vm + q;
         int st = exbp[(size_t)tcur * (Q + 2) + q];
         double exitLike = exl[(size_t)tcur * (Q + 2) + q];
         double nextLike = exitLike;
         int segEndT = tcur;
         if (st == 1) {                                      /* tee pass-through: the model takes no frame */
            double entryLike = entAt[(size_t)tcur * (Q + 2) + q];
            modStart[q - 1] = tcur; modEnd[q - 1] = tcur; modScore[q - 1] = exitLike - entryLike;
            q--;
            continue;
         }
         modEnd[q - 1] = tcur;
         while (1) {
            int p = bp[(size_t)tcur * nSlots + v->slot0 + st - 2];
            if (p != st) {                                   /* state st was entered at frame tcur */
               if (nSeg >= maxSeg) { nSeg = -3; goto done; }
               segQ[nSeg] = q; segState[nSeg] = st; segStart[nSeg] = tcur - 1; segEnd[nSeg] = segEndT;
               segScore[nSeg] = nextLike - pre[(size_t)tcur * nSlots + v->slot0 + st - 2];
               nSeg++;
               nextLike = pre[(size_t)tcur * nSlots + v->slot0 + st - 2];
               segEndT = tcur - 1;
               if (p == 1) {                                 /* from the entry state: model boundary */
                  double entryLike = entAt[(size_t)(tcur - 1) * (Q + 2) + q];
                  modStart[q - 1] = tcur - 1;
                  modScore[q - 1] = exitLike - entryLike;
                  tcur--; q--;
                  break;
               }
               st = p;
            }
            tcur--;
         }
      }
      /* reverse into time order */
      for (i = 0; i < nSeg / 2; i++) {
         int k = nSeg - 1 - i, ti; double td;
         ti = segQ[i]; segQ[i] = segQ[k]; segQ[k] = ti;
         ti = segState[i]; segState[i] = segState[k]; segState[k] = ti;
         ti = segStart[i]; segStart[i] = segStart[k]; segStart[k] = ti;
         ti = segEnd[i]; segEnd[i] = segEnd[k]; segEnd[k] = ti;
         td = segScore[i]; segScore[i] = segScore[k]; segScore[k] = td;
      }
   }
done:
   free(vm); free(like); free(nw); free(exitL); free(entL); free(instMax); free(active);
   free(pre); free(bp); free(exl); free(entAt); free(exbp);
   return nSeg;
}
