// fb_kernels.hip -- K2 (beta pass), K3 (alpha pass + occupation / transition statistics) and
// K4 (mixture statistics) of the embedded Baum-Welch step.
//
// Reference semantics restated here (all of HFB.c, S==1, PLAINHS/SHAREDHS, no xforms):
//   SetBeta 1149-1296, StepAlpha 686-784, InitAlpha 616-651, MaxModelProb 655-682,
//   SetOcct 399-418, UpTranParms 1371-1423, UpMixParms 1426-1744, StepBack retry loop 1321-1366.
//
// MI355X mapping.  The recursions are sequential in time and only as wide as the utterance's
// model chain (Q models x N states, ~200 cells), so ONE WORKGROUP OWNS ONE UTTERANCE and one
// thread owns one (model,state) cell; thousands of utterances are resident at once (1 block each),
// which is where the parallelism comes from.  The alpha/beta columns of the current and previous
// frame live in LDS (doubles, like the reference's DVectors); the transition row/column of each
// cell is staged in LDS once; the full beta trellis goes to HBM as [t][cell] so a column is one
// coalesced 8-byte-per-lane store/load.  Transition and occupation counts are summed over time in
// thread-private LDS slots and flushed with one fp64 atomic per (cell,entry) at the end of the
// utterance (HW global_atomic_add_f64), so a transition matrix shared by every model is not hit
// once per frame.  Mixture-level statistics are not computed inside the sequential loop: K3 writes
// log-occupancy seeds densely ([t][slot], LZERO when below the MINFORPROB prune) and K4 scans them
// with whole waves, recomputing the per-component likelihoods only for the ~2% of (frame,state)
// pairs that survive the prune.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "ladd.h"

// every LAdd of the recursions goes through the LDS table (ladd.h); `ltab` is the block's copy
#define ladd(x, y, mle) ladd_tab((x), (y), (mle), ltab)

// SetOcct/UpTranParms add exp(x) whenever x > MINEARG (-708.3).  exp(-100) = 3.7e-44 is below the smallest float
// increment the reference's float counters can register next to any real count, so the exponential is only
// evaluated above this floor (saves ~30 fp64 instructions per skipped term in the sequential loop).
#define EXPFLOOR (-100.0)

struct CellMeta {
   int q, i, N, mc0, ms0;      // model (1-based), state (1..N), states in model, cell of state 1, slot of state 2
};

struct LdsCarve {
   char *p;
   __device__ explicit LdsCarve(void *base) : p((char *)base) {}
   template <typename T> __device__ T *take(size_t n)
   {
      T *r = (T *)p;
      p += ((n * sizeof(T) + 7) & ~(size_t)7);
      return r;
   }
};

// ------------------------------------------------------------------------------------ K2: beta
__global__ void k_beta(FbArgs a)
{
   const int nCellsMax = a.nCellsMax, QMax = a.QMax;
   extern __shared__ double smem[];
   const int u = a.uttList[blockIdx.x], tid = threadIdx.x;
   const UttDesc ud = a.utt[u];
   if (ud.status != HTKAMD_UTT_OK) {
      if (tid == 0) { a.status[u] = ud.status; a.pr[u] = LZERO; }
      return;
   }
   const int T = ud.T, Q = ud.Q, nC = ud.nCells, maxN = a.maxN;
   LdsCarve lds(smem);
   double *colA = lds.take<double>(nCellsMax);
   double *colB = lds.take<double>(nCellsMax);
   double *maxP = lds.take<double>(QMax + 3);
   float *trow = lds.take<float>((size_t)nCellsMax * maxN);
   float *mA1N = lds.take<float>(QMax + 3);
   int *mC0 = lds.take<int>(QMax + 3);
   int *mNq = lds.take<int>(QMax + 3);
   int *mDm = lds.take<int>(QMax + 3);
   int *sh = lds.take<int>(8);
   double *ltab = lds.take<double>(LADD_TAB_DOUBLES);
   ladd_table_to_lds(ltab, a.laddTab);

   CellMeta cm = {0, 0, 0, 0, 0};
   // threads are grouped by role (entry | emitting | exit cells, each padded to whole waves) so that a wave
   // executes one role's code path; c is this thread's cell in the q-major cell order used by the LDS columns
   const int c = (tid < ud.nThr) ? (int)a.thrCell[ud.thr0 + tid] : -1;
   const bool live = c >= 0;
   if (live) {
      cm.q = a.cQ[ud.cell0 + c]; cm.i = a.cI[ud.cell0 + c];
      const int mi = ud.q0 + cm.q - 1;
      cm.N = a.mN[mi]; cm.mc0 = a.mCell0[mi]; cm.ms0 = a.mSlot0[mi];
      const float *tp = a.transP + a.mTp[mi];
      for (int j = 1; j <= cm.N; j++) trow[c * maxN + (j - 1)] = tp[(cm.i - 1) * cm.N + (j - 1)];
   }
   for (int q = tid + 1; q <= Q; q += blockDim.x) {
      const int mi = ud.q0 + q - 1, N = a.mN[mi];
      mC0[q] = a.mCell0[mi]; mNq[q] = N; mDm[q] = a.mDms[mi];
      mA1N[q] = a.transP[a.mTp[mi] + (N - 1)];          // a_1N: the tee transition
   }
   if (tid == 0) { mC0[Q + 1] = 0; mNq[Q + 1] = 0; mDm[Q + 1] = 1; mA1N[Q + 1] = (float)LZERO; mA1N[0] = (float)LZERO; mDm[0] = 1; }
   __syncthreads();

   const short *tLo = a.taperLo + ud.frame0 - 1, *tHi = a.taperHi + ud.frame0 - 1;   // 1-based t
   short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
   const float *outp = a.outp + ud.outp0;
   double *gbeta = a.beta + ud.beta0;
   const double mle = a.minLogExp;
   const bool pruning = a.pruneInit < HTKAMD_NOPRUNE;

   // HTKAMD_COMPAT_STREAM_REVISIT -- Setotprob(t, hiQ, loQ) for S > 1 as the reference computes it (HFB.c:1015-1066): it walks the models
   // hiQ .. loQ - 1 and a model's states 2 .. N - 1; a tied state it has met before IN THIS CALL gets the float sum of the streams'
   // replaced values (sum - x_s), halved (:1059), the others the sum of the streams.  Every retry of StepBack starts afresh (:559-562),
   // so the state's row is rewritten for every model the call covers.
   const int mySlot = (live && cm.i > 1 && cm.i < cm.N) ? cm.ms0 + cm.i - 2 : -1;
   auto setotprob_revisit = [&](int t, int hiQ, int loQ) {
      if (loQ > 1) --loQ;
      if (mySlot >= 0 && cm.q >= loQ && cm.q <= hiQ) {
         const int nx = a.nextSame[ud.slot0 + mySlot];
         const bool seen = nx >= 0 && (int)a.sQ[ud.slot0 + nx] <= hiQ;
         const float *pu = a.outpU + ud.outp0 * a.NSt + (size_t)mySlot * T + (t - 1);
         const size_t strideK = (size_t)ud.nSlots * T;
         float sum = 0.0f;
         for (int k = 0; k < a.NSt; k++) sum += pu[k * strideK];
         if (seen) {
            float s2 = 0.0f;
            for (int k = 0; k < a.NSt; k++) s2 += sum - pu[k * strideK];
            sum = s2 / 2;
         }
         ((float *)outp)[(size_t)mySlot * T + (t - 1)] = sum;
      }
      __syncthreads();
   };

   double thresh = a.pruneInit, pr = LZERO;
   int ok = 0;
   for (;;) {                                            // StepBack retry loop (HFB.c:1332-1361)
      double *colC = colA, *colN = colB;
      int fail = 0;
      // ---- t = T (HFB.c:1175-1198)
      int endq = tLo[T];
      if (a.compatRevisit) setotprob_revisit(T, Q, endq);
      if (tid == 0) {
         double e = 0.0;
         for (int q = Q; q >= endq; q--) {
            e = (q == Q) ? 0.0 : e + (double)mA1N[q + 1];
            colC[mC0[q] + mNq[q] - 1] = e;
         }
      }
      __syncthreads();
      if (live && cm.q >= endq && cm.i > 1 && cm.i < cm.N)
         colC[c] = (double)trow[c * maxN + cm.N - 1] + colC[cm.mc0 + cm.N - 1];
      __syncthreads();
      if (live && cm.q >= endq && cm.i == 1) {
         double x = LZERO;
         for (int j = 2; j < cm.N; j++) {
            double aa = trow[c * maxN + j - 1], y = colC[cm.mc0 + j - 1];
            if (aa > LSMALL && y > LSMALL)
               x = ladd(x, aa + (double)outp[(size_t)(cm.ms0 + j - 2) * T + (T - 1)] + y, mle);
         }
         colC[c] = x;
      }
      __syncthreads();
      if (live && cm.q >= endq) gbeta[(size_t)(T - 1) * nC + c] = colC[c];
      if (tid == 0) { gLo[T] = (short)endq; gHi[T] = (short)Q; }
      int qHiN = Q, qLoN = endq, lastEnd = endq;

      // ---- t = T-1 .. 1 (HFB.c:1205-1277)
      for (int t = T - 1; t >= 1; t--) {
         const int startq = qHiN;
         endq = (qLoN == 1) ? 1 : ((tLo[t] >= qLoN) ? tLo[t] : qLoN - 1);
         while (endq > 1 && mDm[endq - 1] == 0) endq--;
         { double *tmp = colC; colC = colN; colN = tmp; }
         if (a.compatRevisit) setotprob_revisit(t, startq, endq);
         const bool inRange = live && cm.q >= endq && cm.q <= startq;
         if (inRange && cm.i > 1) {
            const int q = cm.q;
            const bool p1 = (q < Q) && (q + 1 >= qLoN) && (q + 1 <= qHiN);
            double ex = p1 ? colN[mC0[q + 1]] : LZERO;
            if (q < startq) {
               const float a1N = mA1N[q + 1];
               if (a1N > (float)LSMALL) {                // next model is a tee model: same-frame exit of q+1
                  const bool p2 = (q + 1 < Q) && (q + 2 >= qLoN) && (q + 2 <= qHiN);
                  double ex1 = p2 ? colN[mC0[q + 2]] : LZERO;
                  ex = ladd(ex, ex1 + (double)a1N, mle);
               }
            }
            if (cm.i == cm.N)
               colC[c] = ex;
            else {
               double x = (double)trow[c * maxN + cm.N - 1] + ex;
               if (q >= qLoN && q <= qHiN)
                  for (int j = 2; j < cm.N; j++) {
                     double aa = trow[c * maxN + j - 1], y = colN[cm.mc0 + j - 1];
                     if (aa > LSMALL && y > LSMALL)
                        x = ladd(x, aa + (double)outp[(size_t)(cm.ms0 + j - 2) * T + t] + y, mle);   // b_j(t+1)
                  }
               colC[c] = x;
            }
         }
         __syncthreads();
         if (inRange && cm.i == 1) {
            double x = LZERO, lMax = LZERO;
            for (int j = 2; j < cm.N; j++) {
               double aa = trow[c * maxN + j - 1], y = colC[cm.mc0 + j - 1];
               if (y > lMax) lMax = y;
               if (aa > LSMALL && y > LSMALL)
                  x = ladd(x, aa + (double)outp[(size_t)(cm.ms0 + j - 2) * T + (t - 1)] + y, mle);      // b_j(t)
            }
            colC[c] = x;
            maxP[cm.q] = lMax;
         }
         __syncthreads();
         int newHi, newLo;
         if (!pruning) {                                 // only the taper acts (HFB.c:1259-1264)
            newHi = (tHi[t] < startq) ? tHi[t] : startq;
            newLo = endq;
         } else {
            if (tid < 64) {                              // beam pruning (HFB.c:1254-1272)
               double g = LZERO;
               for (int q = endq + tid; q <= startq; q += 64) g = fmax(g, maxP[q]);
               for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_xor(g, o));
               if (tid == 0) {
                  int s = startq, e = endq, f = 0;
                  while (s >= 1 && g - maxP[s] > thresh) --s;
                  while (s >= 1 && tHi[t] < s) --s;
                  if (s < 1) f = 1;
                  else
                     while (g - maxP[e] > thresh) { ++e; if (e > s) { f = 1; break; } }
                  sh[0] = s; sh[1] = e; sh[2] = f;
               }
            }
            __syncthreads();
            newHi = sh[0]; newLo = sh[1];
            if (sh[2]) fail = 1;
            __syncthreads();
         }
         if (fail) break;
         if (inRange) gbeta[(size_t)(t - 1) * nC + c] = colC[c];
         if (tid == 0) { gLo[t] = (short)newLo; gHi[t] = (short)newHi; }
         qHiN = newHi; qLoN = newLo; lastEnd = endq;
      }
      if (!fail) {
         pr = colC[mC0[lastEnd]];                        // utt->pr = bqt[1] of the last model processed
         if (pr > LSMALL) { ok = 1; break; }
      }
      __syncthreads();
      thresh += a.pruneInc;
      if (thresh > a.pruneLim || a.pruneInc == 0.0) break;
   }
   if (tid == 0) {
      a.pr[u] = ok ? pr : LZERO;
      a.status[u] = ok ? HTKAMD_UTT_OK : HTKAMD_UTT_SKIPPED;
   }
}

// ------------------------------------------------------------------------------------ K3: alpha + stats
__global__ void k_alpha(FbArgs a)
{
   const int nCellsMax = a.nCellsMax, QMax = a.QMax;
   extern __shared__ double smem[];
   const int u = a.uttList[blockIdx.x], tid = threadIdx.x;
   const UttDesc ud = a.utt[u];
   if (a.status[u] != HTKAMD_UTT_OK) {                   // skipped in the beta pass (or pre-check)
      if (tid == 0) atomicAdd(a.acc + a.lay.nUttSkipped, 1.0);
      return;
   }
   const int T = ud.T, Q = ud.Q, nC = ud.nCells, maxN = a.maxN;
   LdsCarve lds(smem);
   double *acol0 = lds.take<double>(nCellsMax);
   double *acol1 = lds.take<double>(nCellsMax);
   double *bcol = lds.take<double>((size_t)3 * nCellsMax);     // beta columns t-1, t, t+1 (slot = t % 3)
   double *mmp = lds.take<double>(QMax + 3);
   double *tacc = lds.take<double>((size_t)nCellsMax * (maxN + 1));
   double *ltab = lds.take<double>(LADD_TAB_DOUBLES);
   float *ocol = lds.take<float>((size_t)3 * nCellsMax);       // output probs of columns t-1, t, t+1
   float *trow = lds.take<float>((size_t)nCellsMax * maxN);
   float *tcol = lds.take<float>((size_t)nCellsMax * maxN);
   float *mA1N = lds.take<float>(QMax + 3);
   int *mC0 = lds.take<int>(QMax + 3);
   int *mNq = lds.take<int>(QMax + 3);
   int *mDm = lds.take<int>(QMax + 3);
   short *bLo = lds.take<short>(a.TMax + 3);                   // final beta beam of every frame (1-based t)
   short *bHi = lds.take<short>(a.TMax + 3);
   ladd_table_to_lds(ltab, a.laddTab);

   CellMeta cm = {0, 0, 0, 0, 0};
   const int c = (tid < ud.nThr) ? (int)a.thrCell[ud.thr0 + tid] : -1;     // role-grouped threads, see k_beta
   const bool live = c >= 0;
   int cM = 0, cHmm = 0, cTrans = 0;
   if (live) {
      cm.q = a.cQ[ud.cell0 + c]; cm.i = a.cI[ud.cell0 + c];
      const int mi = ud.q0 + cm.q - 1;
      cm.N = a.mN[mi]; cm.mc0 = a.mCell0[mi]; cm.ms0 = a.mSlot0[mi];
      cHmm = a.mHmm[mi]; cTrans = a.mTrans[mi];
      const float *tp = a.transP + a.mTp[mi];
      for (int j = 1; j <= cm.N; j++) {
         trow[c * maxN + (j - 1)] = tp[(cm.i - 1) * cm.N + (j - 1)];
         tcol[c * maxN + (j - 1)] = tp[(j - 1) * cm.N + (cm.i - 1)];
      }
      for (int j = 0; j <= maxN; j++) tacc[c * (maxN + 1) + j] = 0.0;
      if (cm.i > 1 && cm.i < cm.N) {
         const int s = a.slotState[ud.slot0 + cm.ms0 + cm.i - 2];
         cM = a.stateCompOff[s + 1] - a.stateCompOff[s];
      }
   }
   for (int q = tid + 1; q <= Q; q += blockDim.x) {
      const int mi = ud.q0 + q - 1, N = a.mN[mi];
      mC0[q] = a.mCell0[mi]; mNq[q] = N; mDm[q] = a.mDms[mi];
      mA1N[q] = a.transP[a.mTp[mi] + (N - 1)];
   }
   if (tid == 0) { mC0[Q + 1] = 0; mNq[Q + 1] = 0; mDm[Q + 1] = 1; mA1N[Q + 1] = (float)LZERO; mA1N[0] = (float)LZERO; mDm[0] = 1; mC0[0] = 0; mNq[0] = 0; }
   {
      const short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
      for (int t = tid + 1; t <= T; t += blockDim.x) { bLo[t] = gLo[t]; bHi[t] = gHi[t]; }
      if (tid == 0) { bLo[0] = 1; bHi[0] = 0; bLo[T + 1] = 1; bHi[T + 1] = 0; }
   }

   short *gaLo = a.aLo + ud.frame0 - 1, *gaHi = a.aHi + ud.frame0 - 1;
   const double *gbeta = a.beta + ud.beta0;
   double *gam = a.gam + ud.gam0;
   const double mle = a.minLogExp, pr = a.pr[u];
   const double minF = (double)a.minFrwdP;
   const bool wantMix = (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES)) != 0;
   const bool wantTrans = (a.uFlags & HTKAMD_UPTRANS) != 0;
   const int nSlots = ud.nSlots;
   const bool emitting = live && cm.i > 1 && cm.i < cm.N;
   // this cell's row of the state-major output-probability block (only meaningful for emitting cells)
   const float *orow = a.outp + ud.outp0 + (size_t)(emitting ? cm.ms0 + cm.i - 2 : 0) * T;

   // Columns 1 and 2 of beta / output probabilities go into the rings now; from then on column t+2 is
   // requested from HBM at the top of step t and parked in LDS at its end, so no step waits on memory.
   if (live) {
      bcol[(size_t)1 * nCellsMax + c] = gbeta[c];
      if (emitting) ocol[(size_t)1 * nCellsMax + c] = orow[0];
      if (T >= 2) {
         bcol[(size_t)2 * nCellsMax + c] = gbeta[(size_t)nC + c];
         if (emitting) ocol[(size_t)2 * nCellsMax + c] = orow[1];
      }
   }
   double *aC = acol0, *aP = acol1;
   double occAcc = 0.0;
   double xpre = LZERO;          // log sum_i alpha_i(t-1) a_ij (+ entry term) of this emitting cell, before b_j(t)
   int err = 0;
   __syncthreads();
   int sq = 1, eq = bHi[1];

   // ---- t = 1: InitAlpha (HFB.c:616-651)
   if (tid == 0) {
      double a1 = 0.0;
      for (int q = 1; q <= eq; q++) {
         a1 = (q == 1) ? 0.0 : a1 + (double)mA1N[q - 1];
         aC[mC0[q]] = a1;
      }
   }
   __syncthreads();
   if (emitting) {
      double v = LZERO;
      if (cm.q <= eq) {
         const double aa = tcol[c * maxN + 0];
         xpre = aC[cm.mc0] + aa;
         if (aa > LSMALL) v = xpre + (double)ocol[(size_t)1 * nCellsMax + c];
      }
      aC[c] = v;
   }
   if (live && cm.q > eq && cm.i == 1) aC[c] = LZERO;
   __syncthreads();
   if (live && cm.i == cm.N) {
      double x = LZERO;
      if (cm.q <= eq)
         for (int i = 2; i < cm.N; i++) {
            const double aa = tcol[c * maxN + i - 1];
            if (aa > LSMALL) x = ladd(x, aC[cm.mc0 + i - 1] + aa, mle);
         }
      aC[c] = x;
   }
   __syncthreads();

   for (int t = 1; t <= T; t++) {
      // request column t+2 (consumed at the end of this step)
      double bNext = 0.0; float oNext = 0.0f;
      const bool haveNext = live && (t + 2 <= T);
      if (haveNext) {
         bNext = gbeta[(size_t)(t + 1) * nC + c];
         if (emitting) oNext = orow[t + 1];
      }
      if (t > 1) {
         // ---- alpha beam (HFB.c:699-722).  mmp[q] = MaxModelProb(q, t-1, minq=q) was left in LDS by step t-1;
         // every thread walks the two short loops itself (uniform LDS reads) instead of waiting for one thread.
         const double *bP = bcol + (size_t)((t - 1) % 3) * nCellsMax;
         const int pLo = bLo[t - 1], pHi = bHi[t - 1], cLo = bLo[t], cHi = bHi[t];
         int s = pLo, e = 0, f = 0;
         while (pr - mmp[s] > minF) { ++s; if (s > cHi) { f = 1; break; } }
         if (!f) {
            if (s < cLo) s = cLo;
            e = (pHi < Q) ? pHi + 1 : pHi;
            for (;;) {
               double m = mmp[e];
               // tee predecessors above the start point (MaxModelProb's qx loop, HFB.c:667-672)
               for (int qx = e - 1; qx > s && mA1N[qx] > (float)LSMALL; qx--) {
                  const int qx1 = qx - 1;
                  if (qx1 >= 1 && qx1 >= pLo && qx1 <= pHi) {
                     const int c1 = mC0[qx1] + mNq[qx1] - 1;
                     const double x = aC[c1] + bP[c1];
                     if (x > m) m = x;
                  }
               }
               if (!(pr - m > minF)) break;
               --e;
               if (e < s) { f = 1; break; }
            }
            if (!f) {
               while (e < Q && mDm[e] == 0) e++;
               if (e > cHi) e = cHi;
            }
         }
         if (f) { err = 1; break; }                      // uniform: every thread computed the same f
         sq = s; eq = e;
         { double *tmp = aC; aC = aP; aP = tmp; }
         // ---- alpha column t (HFB.c:729-771)
         if (live) {
            const int q = cm.q;
            if (q < sq || q > eq) {
               if (cm.i < cm.N) aC[c] = LZERO;
            } else if (cm.i < cm.N) {
               double a1;
               if (q == 1) a1 = LZERO;
               else {
                  a1 = aP[mC0[q - 1] + mNq[q - 1] - 1];
                  const float t1N = mA1N[q - 1];
                  if (q > sq && t1N > (float)LSMALL) {   // previous model is a tee model
                     const double a1p = (q - 1 == 1) ? LZERO : aP[mC0[q - 2] + mNq[q - 2] - 1];
                     a1 = ladd(a1, a1p + (double)t1N, mle);
                  }
               }
               if (cm.i == 1) aC[c] = a1;
               else {
                  double aa = tcol[c * maxN + 0];
                  double x = (aa > LSMALL) ? aa + a1 : LZERO;
                  for (int i = 2; i < cm.N; i++) {
                     aa = tcol[c * maxN + i - 1];
                     const double y = aP[cm.mc0 + i - 1];
                     if (aa > LSMALL && y > LSMALL) x = ladd(x, y + aa, mle);
                  }
                  xpre = x;
                  aC[c] = x + (double)ocol[(size_t)(t % 3) * nCellsMax + c];
               }
            }
         }
         __syncthreads();
         if (live && cm.i == cm.N) {
            double x = LZERO;
            if (cm.q >= sq && cm.q <= eq)
               for (int i = 2; i < cm.N; i++) {
                  const double aa = tcol[c * maxN + i - 1], y = aC[cm.mc0 + i - 1];
                  if (aa > LSMALL && y > LSMALL) x = ladd(x, y + aa, mle);
               }
            aC[c] = x;
         }
      }
      // park column t+2 (its ring slot held column t-1, last read by the beam walk above)
      if (haveNext) {
         bcol[(size_t)((t + 2) % 3) * nCellsMax + c] = bNext;
         if (emitting) ocol[(size_t)((t + 2) % 3) * nCellsMax + c] = oNext;
      }
      __syncthreads();
      if (tid == 0) { gaLo[t] = (short)sq; gaHi[t] = (short)eq; }
      if (a.alphaDbg && live) a.alphaDbg[ud.beta0 + (size_t)(t - 1) * nC + c] = aC[c];

      // ---- statistics for column t (HFB.c:1790-1806) and MaxModelProb of column t for the next beam walk
      if (live) {
         const int q = cm.q, i = cm.i, N = cm.N;
         const double *bT = bcol + (size_t)(t % 3) * nCellsMax;
         const double *bT1 = bcol + (size_t)((t + 1) % 3) * nCellsMax;
         const float *oT = ocol + (size_t)(t % 3) * nCellsMax;
         const float *oT1 = ocol + (size_t)((t + 1) % 3) * nCellsMax;
         const int cLo = bLo[t], cHi = bHi[t];
         if (i == 1) {                                   // HFB.c:655-682 with minq == q
            double m = LZERO;
            if (q > 1 && q - 1 >= cLo && q - 1 <= cHi) {
               const int c1 = mC0[q - 1] + mNq[q - 1] - 1;
               m = aC[c1] + bT[c1];
            }
            if (q >= cLo && q <= cHi)
               for (int i2 = 1; i2 < N; i2++) {
                  const double x = aC[cm.mc0 + i2 - 1] + bT[cm.mc0 + i2 - 1];
                  if (x > m) m = x;
               }
            mmp[q] = m;
         }
         const bool inBeam = q >= sq && q <= eq;
         double seed = LZERO;
         if (inBeam) {
            const bool bqt1ok = (t < T) && q >= bLo[t + 1] && q <= bHi[t + 1];
            const bool bq1tok = (q < Q) && (q + 1) >= cLo && (q + 1) <= cHi;
            const double ai = aC[c], bi = bT[c];
            // SetOcct (HFB.c:399-418)
            double x = ai + bi;
            const float a1N = trow[c * maxN + N - 1];
            if (i == 1 && bq1tok && a1N > (float)LSMALL) x = ladd(x, ai + bT[mC0[q + 1]] + (double)a1N, mle);
            x -= pr;
            const float occ = (x > EXPFLOOR) ? (float)exp(x) : 0.0f;
            if (i < N) occAcc += (double)occ;
            if (wantTrans && i < N) {                    // UpTranParms (HFB.c:1390-1410), row i
               double *ta = tacc + c * (maxN + 1);
               if (i == 1) {
                  for (int j = 2; j < N; j++) {
                     x = ai + (double)trow[c * maxN + j - 1] + (double)oT[cm.mc0 + j - 1] + bT[cm.mc0 + j - 1] - pr;
                     if (x > EXPFLOOR) ta[j] += exp(x);
                  }
                  if (a1N > (float)LSMALL && bq1tok) {
                     x = ai + (double)a1N + bT[mC0[q + 1]] - pr;
                     if (x > EXPFLOOR) ta[N] += exp(x);
                  }
               } else {
                  if (bqt1ok)
                     for (int j = 2; j < N; j++) {
                        x = ai + (double)trow[c * maxN + j - 1] + (double)oT1[cm.mc0 + j - 1] + bT1[cm.mc0 + j - 1] - pr;
                        if (x > EXPFLOOR) ta[j] += exp(x);
                     }
                  x = ai + (double)trow[c * maxN + N - 1] + bT[cm.mc0 + N - 1] - pr;
                  if (x > EXPFLOOR) ta[N] += exp(x);
               }
            }
            if (wantMix && i > 1 && i < N) {             // UpMixParms seed (HFB.c:1479-1489,1573-1606)
               if (cM == 1 || a.maxM == 1) {
                  x = ai + bi - pr;
                  if (-x < minF) seed = x;
               } else {
                  // initx = log(a_1j alpha_1(t) + sum_i alpha_i(t-1) a_ij) + beta_j - pr (HFB.c:1481-1488).  The sum is
                  // the one the alpha recursion formed before adding b_j(t) (same operands, same order); the two
                  // only differ by how log-zero terms are skipped, which cannot lift a sum above LSMALL.
                  double initx = xpre;
                  initx += bi - pr;
                  // every component's x = initx + logw + prob is <= initx + b_j(t) (+ float rounding)
                  // (HTKAMD_COMPAT_STREAM_REVISIT: the row may hold a second visit's value, (NS - 1) / 2 times the state's log probability --
                  //  below it for four streams and more, so no bound: k_mixstats_ms weighs every component of the pair itself)
                  const double ub = initx + (double)oT[c];
                  if (ub > -minF - 0.01 || a.compatRevisit) seed = initx;
               }
            }
         }
         if (i > 1 && i < N) gam[(size_t)(t - 1) * nSlots + cm.ms0 + i - 2] = seed;
      }
      __syncthreads();
   }

   if (err) {
      if (tid == 0) { a.status[u] = HTKAMD_UTT_EALPHA; atomicAdd(a.acc + a.lay.nUttSkipped, 1.0); }
      return;
   }
   // ---- flush the per-cell sums.  When every model of the utterance uses the same transition matrix (the usual case), the
   // cells are first summed per (row, column) in LDS: same-address f64 atomics serialise in L2 (see fb_wave.hip).
   if (wantTrans) {
      __shared__ int tShared, tMixed;
      if (tid == 0) { tShared = a.mTrans[ud.q0]; tMixed = 0; }
      __syncthreads();
      if (live && cTrans != tShared) tMixed = 1;
      if (live) tacc[c * (maxN + 1)] = (cm.i < cm.N) ? occAcc : 0.0;       // column 0 of a cell's row is free: its occupation count
      __syncthreads();
      if (!tMixed) {
         const int N0 = a.mN[ud.q0], t0 = tShared;
         for (int idx = tid; idx < maxN * (maxN + 1); idx += blockDim.x) {
            const int i = idx / (maxN + 1) + 1, j = idx % (maxN + 1);
            if (i >= N0 || j == 1 || j > N0) continue;
            double v = 0.0;
            for (int cc = 0; cc < ud.nCells; cc++)
               if (a.cI[ud.cell0 + cc] == i) v += tacc[cc * (maxN + 1) + j];
            if (v == 0.0) continue;
            if (j == 0) atomicAdd(a.acc + a.lay.trOcc + a.trOccOff[t0] + (i - 1), v);
            else atomicAdd(a.acc + a.lay.tr + a.transOff[t0] + (size_t)(i - 1) * N0 + (j - 1), v);
         }
      } else if (live && cm.i < cm.N) {
         const double *ta = tacc + c * (maxN + 1);
         double *tr = a.acc + a.lay.tr + a.transOff[cTrans] + (size_t)(cm.i - 1) * cm.N;
         for (int j = 2; j <= cm.N; j++)
            if (ta[j] != 0.0) atomicAdd(tr + (j - 1), ta[j]);
         if (occAcc != 0.0) atomicAdd(a.acc + a.lay.trOcc + a.trOccOff[cTrans] + (cm.i - 1), occAcc);
      }
   }
   if (live && cm.i == 1) atomicAdd(a.acc + a.lay.nEgs + cHmm, 1.0);
   if (tid == 0) {
      atomicAdd(a.acc + a.lay.totalPr, pr);
      atomicAdd(a.acc + a.lay.totalT, (double)T);
      atomicAdd(a.acc + a.lay.nUttDone, 1.0);
      atomicAdd(a.acc + a.lay.nEval, (double)ud.nEval);
   }
}

// ------------------------------------------------------------------------------------ K4: mixture statistics
// DT > 0: vector size known at compile time (all parameter loads of a component are issued together);
// DT == 0: any size.
//
// mix_hit: UpMixParms (HFB.c:1573-1721) for up to 64/GS surviving (frame, state) pairs side by side -- GS lanes per pair (GS >= the
// largest mixture count, a power of two), lane % GS = component.  Per lane group: `ok`, the tied state `st`, the row of the frame in
// the feature table and the pair's seed (the state's log occupation without the component's own score, or with it for single
// Gaussians).  Called by every lane of the wavefront; recBase / recUsed: the wavefront's block of the record list.
// what follows the posterior of a lane's component (pass, Lr; g = its Gaussian, c0 + m its component, s the tied state): weight counts,
// the record list of the per-Gaussian reduction, direct atomics for what the list has no room for
template <int GS>
__device__ __forceinline__ void mix_post(const FbArgs &a, const int s, const int c0, const int m, const int g, const bool pass, const double Lr,
                                         const int frameRow, const float *xrow, int &recBase, int &recUsed)
{
   const int lane = threadIdx.x & 63, sub = lane % GS;
   const int D = a.D;
   const bool upMu = a.uFlags & HTKAMD_UPMEANS, upVa = a.uFlags & HTKAMD_UPVARS, upWt = a.uFlags & HTKAMD_UPMIXES;
   double sumLr = pass ? Lr : 0.0;
#pragma unroll
   for (int o = GS / 2; o > 0; o >>= 1) sumLr += __shfl_xor(sumLr, o);      // within the hit's lane group
   if (sub == 0 && sumLr != 0.0) atomicAdd(a.acc + a.lay.wtOcc + s, sumLr);
   // record path: list (Gaussian, frame, posterior) for the per-Gaussian reduction (k_rec_reduce).  A wavefront takes list
   // space in blocks of 64 records (one atomic on the shared cursor per block, not per hit: a single hot address serialises
   // in L2); what is left of a block when the next is taken, or at the end, is filled with empty records (g = -1)
   bool stored = false;
   if (a.rec) {
      const unsigned long long pk = __ballot(pass);
      if (pk) {
         const int np = __popcll(pk);
         if (recBase < 0 || recUsed + np > 64) {
            if (recBase >= 0 && recUsed + lane < 64) a.rec[recBase + recUsed + lane].g = -1;
            int base = 0;
            if (lane == 0) base = atomicAdd(a.recCtl, 64);
            base = __shfl(base, 0);
            recBase = (base >= 0 && (long long)base + 64 <= a.recCap) ? base : -1;
            recUsed = 0;
         }
         if (recBase >= 0) {
            if (pass) {
               MixRec r; r.g = g; r.frame = frameRow; r.L = Lr;
               a.rec[recBase + recUsed + __popcll(pk & ((1ull << lane) - 1))] = r;
               atomicAdd(a.recCtl + 1 + g, 1);
               stored = true;
            }
            recUsed += np;
         }
      }
   }
   if (pass) {
      if (upMu && !stored) atomicAdd(a.acc + a.lay.muOcc + g, Lr);
      if (upVa && !stored) atomicAdd(a.acc + a.lay.vaOcc + g, Lr);
      if (upWt) atomicAdd(a.acc + a.lay.wt + c0 + m, Lr);
   }
   // first-order statistics of every surviving (hit, component): the whole wave, lane = dimension
   unsigned long long pm = __ballot(pass && !stored);        // what the list had no room for: direct atomics
   while (pm) {
      const int ml = __ffsll((long long)pm) - 1;
      pm &= pm - 1;
      const double L = __shfl(Lr, ml);
      const int gg = __shfl(g, ml);
      const unsigned long long xp = (unsigned long long)xrow;
      const float *xr = (const float *)(((unsigned long long)__shfl((int)(xp >> 32), ml) << 32) | (unsigned int)__shfl((int)(xp & 0xffffffffu), ml));
      const float *mean = a.mean + (size_t)gg * D;
      for (int k = lane; k < D; k += 64) {
         const float z = xr[k] - mean[k];
         if (upMu && upVa) {                    // HFB.c:1673-1678
            const float zl = (float)((double)z * L);
            atomicAdd(a.acc + a.lay.mu + (size_t)gg * D + k, (double)zl);
            atomicAdd(a.acc + a.lay.va + (size_t)gg * D + k, (double)(z * zl));
         } else if (upMu) {                     // HFB.c:1697-1698
            atomicAdd(a.acc + a.lay.mu + (size_t)gg * D + k, (double)z * L);
         } else if (upVa) {                     // HFB.c:1706-1709
            atomicAdd(a.acc + a.lay.va + (size_t)gg * D + k, (double)(z * z) * L);
         }
      }
   }
}

template <int DT, int GS>
__device__ __forceinline__ void mix_hit(const FbArgs &a, const bool ok, const int st, const int frameRow, const double seed, int &recBase, int &recUsed)
{
   const int lane = threadIdx.x & 63, sub = lane % GS;
   const int D = DT > 0 ? DT : a.D;
   const double minF = (double)a.minFrwdP;
   const int s = ok ? st : 0;
   const int c0 = a.stateCompOff[s], M = ok ? a.stateCompOff[s + 1] - c0 : 0;
   const float *xrow = a.X + (size_t)(ok ? frameRow : 0) * D;
   // per-component posterior (lane%GS = component): x = initx + logw + prob (HFB.c:1581-1606)
   int Mmax = M;
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(Mmax, o); Mmax = w > Mmax ? w : Mmax; }
   for (int mb = 0; mb < Mmax; mb += GS) {
      const int m = mb + sub;
      bool pass = false;
      double Lr = 0.0;
      int g = 0;
      if (m < M) {
         g = a.compGauss[c0 + m];
         if (M == 1 || a.maxM == 1) { pass = true; Lr = exp(seed); }
         else {
            const float wt = a.compLogWt[c0 + m];
            if (wt > (float)LMINMIX) {
               const float *P = a.gparam + (size_t)g * a.PS;
               float sum = P[2 * D];
               if (DT > 0) {
                  // (mean, ivar) pairs, two per 16-byte load (rows are PS = 4k floats long and 16-byte aligned): a lane reads its
                  // own 320-byte row, so every load instruction touches one cache line per lane -- half as many instructions, half
                  // as many L1 tag look-ups as with 8-byte loads (the kernel was bound by those: ~620 line touches per hit)
                  const float4 *P4 = (const float4 *)P;
                  // in batches of 4 loads (8 dimensions): the kernel is a chain of dependent loads per hit, hidden only by other
                  // wavefronts -- 180 VGPRs (all 39 pairs in flight) left room for 2 per SIMD
                  constexpr int NQ = (DT + 1) / 2;
#pragma unroll 1
                  for (int q0 = 0; q0 < NQ; q0 += 4) {
                     float4 pv[4]; float xv[8];
#pragma unroll
                     for (int i = 0; i < 4; i++) if (q0 + i < NQ) pv[i] = P4[q0 + i];
#pragma unroll
                     for (int i = 0; i < 8; i++) if (2 * q0 + i < DT) xv[i] = xrow[2 * q0 + i];
#pragma unroll
                     for (int i = 0; i < 8; i++)
                        if (2 * q0 + i < DT) {
                           const float mu = (i & 1) ? pv[i >> 1].z : pv[i >> 1].x, iv = (i & 1) ? pv[i >> 1].w : pv[i >> 1].y;
                           const float xmm = xv[i] - mu;
                           sum += xmm * xmm * iv;
                        }
                  }
               } else {
                  for (int i = 0; i < D; i++) {
                     const float xmm = xrow[i] - P[2 * i];
                     sum += xmm * xmm * P[2 * i + 1];
                  }
               }
               const float prob = -0.5f * sum;
               const double x = (seed + (double)wt) + (double)prob;
               if (-x < minF) { pass = true; Lr = exp(x); }
            }
         }
      }
      mix_post<GS>(a, s, c0, m, g, pass, Lr, frameRow, xrow, recBase, recUsed);
   }
}

// k_mixstats: the pairs come from the DENSE seed array gam[u][t][slot] the alpha kernels of fb_state.hip / fb_wave.hip / this file write
// (log-zero where the MINFORPROB prune lets nothing through: ~98 % of it)
template <int DT, int GS>
__global__ __launch_bounds__(256, 4) void k_mixstats(FbArgs a)
{
   constexpr int HPS = 64 / GS;
   __shared__ unsigned short hitIdx[4][512];
   __shared__ double hitSeed[4][512];
   const int lane = threadIdx.x & 63;
   const int grp = lane / GS;
   const size_t nWaves = ((size_t)gridDim.x * blockDim.x) >> 6;
   const size_t waveId = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   int recBase = -1, recUsed = 0;                        // this wavefront's current block of the record list
   // dense scan of the seed array: 8 x 64 seeds per wave and iteration (8 independent 512-byte loads in flight)
   for (size_t base0 = waveId * 512; base0 < a.gamTotal; base0 += nWaves * 512) {
    double vv[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
       const size_t idx = base0 + (size_t)r * 64 + lane;
       vv[r] = (idx < a.gamTotal) ? a.gam[idx] : LZERO;
    }
    const int cu = a.gamChunkUtt[base0 >> 9];            // utterance of the chunk's first seed (host table)
    // hits are sparse (a few per 512 seeds): collect them in a wave-private LDS list first, so that the lane groups
    // always have HPS hits to work on side by side
    volatile unsigned short *hIdx = hitIdx[threadIdx.x >> 6];
    volatile double *hSeed = hitSeed[threadIdx.x >> 6];
    int count = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
       const double v = vv[r];
       const unsigned long long hm = __ballot(v > LSMALL);
       if (v > LSMALL) {
          const int pos = count + __popcll(hm & ((1ull << lane) - 1));
          hIdx[pos] = (unsigned short)(r * 64 + lane); hSeed[pos] = v;
       }
       count += __popcll(hm);
    }
    for (int i0 = 0; i0 < count; i0 += HPS) {
       const int src = (i0 + grp < count) ? i0 + grp : -1;
       const bool have = src >= 0;
       const size_t hidx = base0 + (have ? hIdx[src] : 0);
       const double seed = have ? hSeed[src] : LZERO;
       // utterance of this entry: advance from the chunk's first utterance (seeds are laid out utterance by utterance)
       int u = cu;
       while (u + 1 < a.nUtt && a.gamOffByUtt[u + 1] <= hidx) u++;
       const UttDesc *up = a.utt + u;                      // only four fields of the descriptor are needed
       const size_t udGam0 = up->gam0;
       const int udSlots = up->nSlots, udSlot0 = up->slot0, udFrame0 = up->frame0;
       // utterances of the left-to-right path have no seeds here (their statistics kernel lists the pairs itself: k_mixhits)
       const bool ok = have && a.status[u] == HTKAMD_UTT_OK && up->pad != 2;
       const size_t rel = hidx - udGam0;
       const int nSl = udSlots > 0 ? udSlots : 1;
       const int t0 = (int)(rel / nSl), slot = (int)(rel % nSl);
       mix_hit<DT, GS>(a, ok, ok ? a.slotState[udSlot0 + slot] : 0, udFrame0 + (ok ? t0 : 0), seed, recBase, recUsed);
    }
   }
   if (recBase >= 0 && recUsed + lane < 64) a.rec[recBase + recUsed + lane].g = -1;
}

// a wave-uniform double held in scalar registers (the compiler keeps a converted kernel argument in a vector pair otherwise)
__device__ __forceinline__ double uniform_f64(double v)
{
   const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
   const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
   return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// k_mixhits: the pairs come as LISTS (fb_lr.hip: every wavefront of k_stats_lr owns a region of the list -- room for all its (frame,
// state) pairs, so nothing can overflow and no cursor is shared -- and leaves the number of 16-byte records it wrote in hitCtl):
// ~20 MB instead of the 0.5 GB seed array written and read back at the bench workload
template <int DT, int GS>
__global__ __launch_bounds__(256, 3) void k_mixhits(FbArgs a)
{
   constexpr int HPS = 64 / GS;
   __shared__ int hitSt[4][64], hitFrame[4][64];
   __shared__ double hitSeed[4][64];
   __shared__ unsigned long long hitKey[4][64];
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int grp = lane / GS, sub = lane % GS;
   const int nWaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
   const int waveId = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
   if (a.stCnt && a.stBucket && !a.hitSlots && a.stCnt[a.nTiedStates] == 0) return;      // every pair found room in its state's bucket (k_mixstate)
   int recBase = -1, recUsed = 0;
   volatile int *hSt = hitSt[wv], *hFr = hitFrame[wv];
   volatile double *hSeed = hitSeed[wv];
   volatile unsigned long long *hKey = hitKey[wv];
   for (int b = waveId; b < a.nHitRegions; b += nWaves) {
      const int cnt = a.hitCtl[b];
      const MixHit *reg = a.hits + (size_t)b * a.hitRegionCap;
      for (int c0r = 0; c0r < cnt; c0r += 64) {
         const int n = (cnt - c0r < 64) ? cnt - c0r : 64;
         if constexpr (DT > 0) {
            // The pairs of a region are a few states over a run of frames.  Sorted by state, a lane group meets the same state several
            // times in a row and keeps its component's (mean, 1/variance) row IN REGISTERS across them: the posterior evaluation read
            // 16 x 320 bytes of parameters per pair (6 GB from L2 / the Infinity Cache per pass at the bench workload), now once per run.
            MixHit h; h.st = 0; h.frame = 0; h.seed = LZERO;
            if (lane < n) h = reg[c0r + lane];
            const unsigned long long key = ((unsigned long long)(unsigned int)h.st << 32) | (unsigned int)h.frame;
            hKey[lane] = (lane < n) ? key : ~0ull;
            int rank = 0;
            for (int jx = 0; jx < n; jx++) { const unsigned long long kj = hKey[jx]; rank += (kj < key || (kj == key && jx < lane)) ? 1 : 0; }
            if (lane < n) { hSt[rank] = h.st; hFr[rank] = h.frame; hSeed[rank] = h.seed; }
            const int per = (n + HPS - 1) / HPS;
            const int i0g = grp * per, i1g = (i0g + per < n) ? i0g + per : n;
            constexpr int NQ = (DT + 1) / 2;
            float4 pm[NQ];
            float gcst = 0.0f, wt = 0.0f;
            int stPrev = -1, c0 = 0, M = 0, g = 0;
            const double minF = uniform_f64((double)a.minFrwdP);      // (in scalar registers: as a vector pair it was SPILLED and reloaded -- with a full vmcnt wait -- in every iteration below)
            for (int it = 0; it < per; it++) {
               const int idx = i0g + it;
               const bool have = idx < i1g;
               const int st = have ? hSt[idx] : 0, frameRow = have ? hFr[idx] : 0;
               const double seed = have ? hSeed[idx] : LZERO;
               if (have && st != stPrev) {
                  stPrev = st;
                  c0 = a.stateCompOff[st]; M = a.stateCompOff[st + 1] - c0;
                  if (sub < M) {
                     g = a.compGauss[c0 + sub];
                     wt = a.compLogWt[c0 + sub];
                     const float4 *P4 = (const float4 *)(a.gparam + (size_t)g * a.PS);
#pragma unroll
                     for (int q = 0; q < NQ; q++) pm[q] = P4[q];
                     gcst = a.gparam[(size_t)g * a.PS + 2 * DT];
                  }
               }
               const float *xrow = a.X + (size_t)frameRow * DT;
               bool pass = false;
               double Lr = 0.0;
               if (have && sub < M) {
                  if (M == 1 || a.maxM == 1) { pass = true; Lr = exp(seed); }
                  else if (wt > (float)LMINMIX) {
                     float sum = gcst;
#pragma unroll
                     for (int q0 = 0; q0 < NQ; q0 += 4) {
                        float xv[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) if (2 * q0 + i < DT) xv[i] = xrow[2 * q0 + i];
#pragma unroll
                        for (int i = 0; i < 8; i++)
                           if (2 * q0 + i < DT) {
                              const float4 pv = pm[q0 + (i >> 1)];
                              const float mu = (i & 1) ? pv.z : pv.x, iv = (i & 1) ? pv.w : pv.y;
                              const float xmm = xv[i] - mu;
                              sum += xmm * xmm * iv;
                           }
                     }
                     const float prob = -0.5f * sum;
                     const double x = (seed + (double)wt) + (double)prob;
                     if (-x < minF) { pass = true; Lr = exp(x); }
                  }
               }
               mix_post<GS>(a, st, c0, sub, g, pass, Lr, frameRow, xrow, recBase, recUsed);
            }
         } else {
            if (lane < n) { const MixHit h = reg[c0r + lane]; hSt[lane] = h.st; hFr[lane] = h.frame; hSeed[lane] = h.seed; }
            for (int i0 = 0; i0 < n; i0 += HPS) {
               const int src = (i0 + grp < n) ? i0 + grp : -1;
               const bool have = src >= 0;
               mix_hit<DT, GS>(a, have, have ? hSt[src] : 0, have ? hFr[src] : 0, have ? hSeed[src] : LZERO, recBase, recUsed);
            }
         }
      }
   }
   if (recBase >= 0 && recUsed + lane < 64) a.rec[recBase + recUsed + lane].g = -1;
}

// ---- record path of K4: bucket the records by Gaussian (counting sort: counts came with the records), then one wavefront per
// Gaussian sums its records' first- and second-order terms (lane = dimension) and adds them to the accumulators ONCE.
#define REC_TILE 2048
__global__ __launch_bounds__(256) void k_rec_tilesum(int *ctl, int G, int *tileSum)
{
   __shared__ int part[4];
   if (ctl[0] == 0) return;                              // no record was listed (k_mixstate took every pair): the four kernels have nothing to do
   const int *cnt = ctl + 1;
   int sum = 0;
   for (int i = blockIdx.x * REC_TILE + threadIdx.x; i < (blockIdx.x + 1) * REC_TILE && i < G; i += 256) sum += cnt[i];
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
   if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
   __syncthreads();
   if (threadIdx.x == 0) tileSum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// exclusive scan of the counts: every block adds up the tiles before its own (a few dozen), then scans its tile (8 per thread)
__global__ __launch_bounds__(256) void k_rec_tilescan(int *ctl, int G, const int *tileSum)
{
   __shared__ int wsum[4];
   __shared__ int offSh;
   if (ctl[0] == 0) return;
   const int *cnt = ctl + 1;
   int *start = ctl + 1 + (G + 1);
   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   int off = 0;
   for (int b = tid; b < (int)blockIdx.x; b += 256) off += tileSum[b];
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o);
   if (lane == 0) wsum[wv] = off;
   __syncthreads();
   if (tid == 0) offSh = wsum[0] + wsum[1] + wsum[2] + wsum[3];
   __syncthreads();
   off = offSh;
   __syncthreads();
   const int i0 = blockIdx.x * REC_TILE + tid * 8;
   int v[8], mine = 0;
#pragma unroll
   for (int k = 0; k < 8; k++) { v[k] = (i0 + k < G) ? cnt[i0 + k] : 0; mine += v[k]; }
   int inc = mine;                                       // inclusive scan over the wave, then over the four waves
#pragma unroll
   for (int o = 1; o < 64; o <<= 1) { const int w = __shfl_up(inc, o); if (lane >= o) inc += w; }
   if (lane == 63) wsum[wv] = inc;
   __syncthreads();
   int run = off + inc - mine;
   for (int w = 0; w < wv; w++) run += wsum[w];
#pragma unroll
   for (int k = 0; k < 8; k++) { if (i0 + k < G) start[i0 + k] = run; run += v[k]; }
   if (i0 <= G - 1 && G - 1 < i0 + 8) start[G] = run;
}

__global__ __launch_bounds__(256) void k_rec_scatter(FbArgs a)
{
   const int G = a.G;
   if (a.recCtl[0] == 0) return;
   const int *start = a.recCtl + 1 + (G + 1);
   int *cur = a.recCtl + 1 + 2 * (G + 1);
   const int n = (a.recCtl[0] >= 0 && a.recCtl[0] < a.recCap) ? a.recCtl[0] : (a.recCap & ~63);
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const MixRec r = a.rec[i];
      if (r.g < 0) continue;
      a.recSorted[start[r.g] + atomicAdd(cur + r.g, 1)] = r;
   }
}

__global__ __launch_bounds__(256) void k_rec_reduce(FbArgs a)
{
   const int G = a.G, D = a.D, lane = threadIdx.x & 63;
   if (a.recCtl[0] == 0) return;
   const int *start = a.recCtl + 1 + (G + 1);
   const bool upMu = a.uFlags & HTKAMD_UPMEANS, upVa = a.uFlags & HTKAMD_UPVARS;
   const int nWaves = (gridDim.x * blockDim.x) >> 6;
   for (int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < G; g += nWaves) {
      const int r0 = start[g], r1 = start[g + 1];
      if (r0 == r1) continue;
      for (int k0 = 0; k0 < D; k0 += 64) {
         const int k = k0 + lane;
         const float mean = (k < D) ? a.mean[(size_t)g * D + k] : 0.0f;
         double sMu = 0.0, sVa = 0.0, sL = 0.0;
         for (int rb = r0; rb < r1; rb += 64) {
            const int nb = (r1 - rb < 64) ? r1 - rb : 64;
            MixRec mine; mine.frame = 0; mine.L = 0.0;
            if (lane < nb) mine = a.recSorted[rb + lane];
            sL += mine.L;
            for (int i = 0; i < nb; i += 4) {                     // four rows in flight
               float xv[4]; double Lv[4];
#pragma unroll
               for (int e = 0; e < 4; e++) {
                  const int ii = (i + e < nb) ? i + e : nb - 1;
                  const int fr = __shfl(mine.frame, ii);
                  Lv[e] = (i + e < nb) ? __shfl(mine.L, ii) : 0.0;
                  xv[e] = (k < D) ? a.X[(size_t)fr * D + k] : 0.0f;
               }
#pragma unroll
               for (int e = 0; e < 4; e++) {
                  if (i + e >= nb) break;
                  const float z = xv[e] - mean;
                  const double L = Lv[e];
                  if (upMu && upVa) {                    // HFB.c:1673-1678
                     const float zl = (float)((double)z * L);
                     sMu += (double)zl; sVa += (double)(z * zl);
                  } else if (upMu) sMu += (double)z * L;           // HFB.c:1697-1698
                  else if (upVa) sVa += (double)(z * z) * L;       // HFB.c:1706-1709
               }
            }
         }
         if (k < D) {
            if (upMu && sMu != 0.0) atomicAdd(a.acc + a.lay.mu + (size_t)g * D + k, sMu);
            if (upVa && sVa != 0.0) atomicAdd(a.acc + a.lay.va + (size_t)g * D + k, sVa);
         }
         if (k0 == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sL += __shfl_xor(sL, o);
            if (lane == 0) {
               if (upMu) atomicAdd(a.acc + a.lay.muOcc + g, sL);
               if (upVa) atomicAdd(a.acc + a.lay.vaOcc + g, sL);
            }
         }
      }
   }
}

// ---- K4s (round 5): mixture statistics bucketed by TIED STATE (left-to-right path, one stream, <= 16 components per state).
// The list kernel above meets a state for ~4 frames in a row (a state lasts that long on the alignment), so it reloads the state's 16
// parameter rows every few pairs -- a chain of three dependent loads each time -- and hands every posterior to a record list that four
// more kernels bucket by Gaussian.  Bucketed by state first (k_stats_sp counts the pairs per state, k_hit_scan / k_hit_scatter make
// the list), ONE wavefront takes ALL pairs of a state: parameters in registers once, the frames' rows through LDS 64 at a time, the
// posteriors of a chunk in LDS, and then the same wavefront with lane = dimension sums mean / variance statistics of the state's
// Gaussians in registers -- one atomic per accumulator element and state at the end, no record list.  UpMixParms HFB.c:1573-1721.
#ifndef LR_EXP_BUILD
#define LR_EXP_BUILD 0
#endif
#define MS_EXP(bit) (LR_EXP_BUILD && (a.lrExp & (bit)))      // ablations of a diagnostic build (tools/lr_exp.py): 256 no sums, 512 fp32 exp, 1024 no rows, 2048 no distances, 8192 no atomics
#ifndef MS_EU
#define MS_EU 8                  /* the launch bound's second figure: 1, 2, 4, 5, 6, 8 give the same 155-register kernel at 0.28-0.29 ms, 3 a 151-register one at 0.33 (tools/r05_var.sh) */
#endif
template <int DT, int MODE>      // MODE: 3 means and variances (HFB.c:1673-1678), 1 means only (:1697), 2 variances only (:1706), 0 weights only
__global__ __launch_bounds__(64, MS_EU) void k_mixstate(FbArgs a)
{
#ifndef MS_CHUNK
#define MS_CHUNK 32
#endif
#ifndef MS_CAP
#define MS_CAP 128               /* pairs per part of a state: 64 / 96 / 128 / 256 give 0.255 / 0.237 / 0.239 / 0.242 ms against 0.285 for whole states (tools/r05_var.sh) */
#endif
#ifndef MS_PARTS
#define MS_PARTS 4
#endif
   constexpr int GS = 16, HPS = 64 / GS, NQ = (DT + 1) / 2, XS = (DT + 3) & ~3;      // XS: floats per row in LDS (16-byte multiple)
   constexpr int CH = MS_CHUNK;                          // pairs per chunk: 14 KB of LDS at 32 (11 workgroups per CU), 24 KB at 64 (6)
   __shared__ float xs[CH * XS];
   __shared__ double Lt[CH][GS];
   __shared__ double hSeed[CH];
   const int st = a.stBase + blockIdx.x, lane = threadIdx.x, grp = lane / GS, sub = lane % GS;
   const int nst = a.stCnt[st] < a.stCap ? a.stCnt[st] : a.stCap;
   // A state's pairs in PARTS of MS_CAP, a wavefront each (blockIdx.y; the last part takes what is left).  Round 6: the kernel's time was its LONGEST
   // wavefront's -- 435 pairs = 14 chunks on the headline set where the mean is 125 = 4.4 chunks (a tied state is used by 1 + Poisson(2.6) models) --
   // 284 us for 133 us worth of work per slot.  A part has its own sums and its own atomics (7 282 parts instead of 5 000 wavefronts at 128).
   const int part = blockIdx.y;
   const int r0 = part * MS_CAP, r1 = (part == (int)gridDim.y - 1 || r0 + MS_CAP > nst) ? nst : r0 + MS_CAP;
   if (r0 >= nst) return;
   const HitS *bucket = a.stBucket + (size_t)st * a.stCap;
   const int c0 = a.stateCompOff[st], M = a.stateCompOff[st + 1] - c0;
   const bool single = (M == 1 || a.maxM == 1);
   constexpr bool upMu = (MODE & 1) != 0, upVa = (MODE & 2) != 0;
   const bool upWt = a.uFlags & HTKAMD_UPMIXES;
   // the state's parameters as pl[dimension pair][component] = (mean d, mean d+1, 1/variance d, 1/variance d+1): the four lane groups of a
   // wavefront work on the same state, so a read is 16 consecutive 16-byte words broadcast to all groups, and two dimensions go through
   // one packed subtract and two packed multiplies (in registers -- 80 of them -- the kernel ran one wavefront per SIMD)
   __shared__ float4 pl[NQ][GS];
   int g = 0;
   float wt = 0.0f, gcst = 0.0f;
   if (sub < M) {
      g = a.compGauss[c0 + sub];
      wt = a.compLogWt[c0 + sub];
      gcst = a.gparam[(size_t)g * a.PS + 2 * DT];
      if (grp == 0) {
         const float4 *P4 = (const float4 *)(a.gparam + (size_t)g * a.PS);
#pragma unroll
         for (int q = 0; q < NQ; q++) { const float4 pv = P4[q]; pl[q][sub] = make_float4(pv.x, pv.z, pv.y, pv.w); }
      }
   }
   // second view of the wavefront: lane = dimension; the means of the state's Gaussians at that dimension
   float meanR[GS];
   // (the Gaussians' numbers by ONE load, a lane each: read one by one they were sixteen scalar loads, each waited for before its mean's load)
   const int gOfLane = (lane < M) ? a.compGauss[c0 + lane] : 0;
#pragma unroll
   for (int m = 0; m < GS; m++) { const int gm_ = __builtin_amdgcn_readlane(gOfLane, m); meanR[m] = (m < M && lane < DT) ? a.mean[(size_t)gm_ * DT + lane] : 0.0f; }
   double sMu[GS], sVa[GS];
#pragma unroll
   for (int m = 0; m < GS; m++) { sMu[m] = 0.0; sVa[m] = 0.0; }
   double wtAcc = 0.0;                                   // posteriors of the lane's component over the pairs of its lane group
   int nTriples = 0;                                     // (frame, state, component) triples that passed, this lane's share
   const double minF = uniform_f64((double)a.minFrwdP);

   for (int base = r0; base < r1; base += CH) {
      const int n = (r1 - base < CH) ? r1 - base : CH;
      __syncthreads();
      int hfr = 0;
      if (lane < n) { const HitS h = bucket[base + lane]; hfr = h.frame; hSeed[lane] = h.seed; }
      // the frames' rows by LDS-DMA, a row per load instruction (its DT lanes read 4 DT contiguous bytes -- a lane fetching the row of its OWN
      // pair touched 64 cache lines per instruction and cost the kernel 0.19 ms): no registers in between, every row of the chunk in flight
      // (the frame numbers are taken with every lane switched on -- lane r of `hfr` is read for r < n <= 32, and at D = 13 / 26 some of those
      //  lanes are outside the loads' `lane < DT` -- only the load itself is predicated)
      // (round 6: the next chunk's rows requested while this chunk is worked on -- two halves of `xs`, the entries a chunk further ahead -- measured:
      //  0.235 ms at 16 pairs per chunk against 0.239 without at 32; the second half's LDS costs the occupancy what the overlap gains.  Not kept.)
      for (int r = 0; r < n; r++) {
         const int fr = MS_EXP(1024) ? 0 : __builtin_amdgcn_readlane(hfr, r);
         if (lane < DT)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.X + (size_t)fr * DT + lane),
                                             (__attribute__((address_space(3))) void *)(xs + r * XS), 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // ---- the posteriors of the chunk: lane group = pair, lane in the group = component (HFB.c:1581-1606)
      for (int it = 0; it * HPS < n; it++) {
         const int h = it * HPS + grp;
         const bool have = h < n;
         bool pass = false;
         double Lr = 0.0;
         if (have && sub < M) {
            const double seed = hSeed[h];
            if (single) { pass = true; Lr = exp(seed); }
            else if (wt > (float)LMINMIX) {
               typedef float v2f __attribute__((ext_vector_type(2)));
               const v2f *xr = (const v2f *)(xs + h * XS);
               float sum = gcst;
               if (!MS_EXP(2048))
#pragma unroll
               for (int q = 0; q < NQ; q++) {                // the reference's sum, a dimension after the other (HModel.c:5420-5431); the products two at a time
                  const float4 pv = pl[q][sub];
                  const v2f xv = xr[q];
                  v2f mu; mu[0] = pv.x; mu[1] = pv.y;
                  v2f iv; iv[0] = pv.z; iv[1] = pv.w;
                  const v2f xmm = xv - mu;
                  const v2f tt = xmm * xmm * iv;
                  sum += tt[0];
                  if (2 * q + 1 < DT) sum += tt[1];
               }
               const float prob = MS_EXP(2048) ? -1.0e30f : -0.5f * sum;
               const double x = (seed + (double)wt) + (double)prob;
               if (-x < minF) { pass = true; Lr = a.fastMath ? (double)__builtin_amdgcn_exp2f((float)x * 1.44269504088896341f) : exp(x); }
            }
         }
         if (have) Lt[h][sub] = pass ? Lr : 0.0;
         wtAcc += pass ? Lr : 0.0;
         nTriples += pass ? 1 : 0;
      }
      __syncthreads();
      // ---- first- and second-order sums of the chunk: lane = dimension, a Gaussian after the other, the pairs it survived in (HFB.c:1673-1709)
      if ((upMu || upVa) && !MS_EXP(256)) {
         // (the sixteen ballots first, four LDS reads in flight at a time: read Gaussian by Gaussian, every one was waited for on its own)
         unsigned long long bmAll[GS];
#pragma unroll
         for (int m4 = 0; m4 < GS; m4 += 4) {
            double Lq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) Lq[i] = (lane < n) ? Lt[lane][m4 + i] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) bmAll[m4 + i] = __ballot(Lq[i] != 0.0);
         }
#pragma unroll
         for (int m = 0; m < GS; m++) {
            if (m < M) {
               unsigned long long bm = bmAll[m];
               // (the running sums of THIS Gaussian as two scalars around the loop, and no branch inside it: updated as elements of the
               // arrays under control flow, every pair cost a copy of all 32 accumulator registers)
               double mu = sMu[m], va = sVa[m];
               const float mean = meanR[m];
               while (bm) {                                  // four pairs' operands requested together; an empty slot adds zeros
                  double Lv[4]; float xv[4];
#pragma unroll
                  for (int i = 0; i < 4; i++) {
                     const bool live = bm != 0;
                     const int h = live ? (int)__builtin_ctzll(bm) : 0;
                     bm &= bm - 1;
                     Lv[i] = live ? Lt[h][m] : 0.0;
                     xv[i] = xs[h * XS + (lane < DT ? lane : 0)];
                  }
#pragma unroll
                  for (int i = 0; i < 4; i++) {
                     const double L = Lv[i];
                     const float z = xv[i] - mean;
                     if constexpr (upMu && upVa) { const float zl = (float)((double)z * L); mu += (double)zl; va += (double)(z * zl); }
                     else if constexpr (upMu) mu += (double)z * L;
                     else va += (double)(z * z) * L;
                  }
               }
               sMu[m] = mu; sVa[m] = va;
            }
         }
      }
   }
   // ---- out: one atomic per accumulator element of the state's Gaussians, component, and the state
   if (MS_EXP(8192)) return;                             // (diagnostic: without the atomics)
#pragma unroll
   for (int m = 0; m < GS; m++) {
      if (m < M && lane < DT) {
         const int gm = a.compGauss[c0 + m];
         if (upMu && sMu[m] != 0.0) atomicAdd(a.acc + a.lay.mu + (size_t)gm * DT + lane, sMu[m]);
         if (upVa && sVa[m] != 0.0) atomicAdd(a.acc + a.lay.va + (size_t)gm * DT + lane, sVa[m]);
      }
   }
   double w = wtAcc;
   w += __shfl_xor(w, 16); w += __shfl_xor(w, 32);       // the four lane groups' shares of the component
   if (grp == 0 && sub < M && w != 0.0) {
      if (upWt) atomicAdd(a.acc + a.lay.wt + c0 + sub, w);
      if (upMu) atomicAdd(a.acc + a.lay.muOcc + g, w);
      if (upVa) atomicAdd(a.acc + a.lay.vaOcc + g, w);
   }
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) nTriples += __shfl_xor(nTriples, o);
   // the state's counts in its own two cells (htkamd_fb_mix_counts sums them): 10 000 updates of two shared cells were 10 us of the kernel's tail
   if (lane == 0) { if (part == 0) a.stCnt[a.nTiedStates + 4 + 2 * st] = nst; atomicAdd(a.stCnt + a.nTiedStates + 5 + 2 * st, nTriples); }      // (zeroed with the buckets' counters before the pass)
   double so = (sub < M) ? w : 0.0;
#pragma unroll
   for (int o = GS / 2; o > 0; o >>= 1) so += __shfl_xor(so, o);
   if (lane == 0 && so != 0.0) atomicAdd(a.acc + a.lay.wtOcc + st, so);
}

// the pass's surviving pairs by tied state (the buckets k_stats_sp filled): a wavefront per state
bool htkamd_mixstate_applies(const FbArgs &a) { return a.stCnt && a.stBucket && !a.hitSlots && a.maxM <= 16 && (a.D == 39 || a.D == 26 || a.D == 13); }
int htkamd_launch_mixstate_range(const FbArgs &a_in, int st0, int st1, hipStream_t s)
{
   if (st0 < 0 || st1 > a_in.nTiedStates || st0 > st1) { htkamd_set_error("mixture statistics: state range [%d, %d) outside the set's %d", st0, st1, a_in.nTiedStates); return HTKAMD_EINVAL; }
   if (st0 == st1) return HTKAMD_OK;
   FbArgs a = a_in;
   a.stBase = st0;
   const int S = st1 - st0;
   const int mode = ((a.uFlags & HTKAMD_UPMEANS) ? 1 : 0) | ((a.uFlags & HTKAMD_UPVARS) ? 2 : 0);
#define MS_LAUNCH(DT_) \
   do { \
      if (mode == 3) hipLaunchKernelGGL((k_mixstate<DT_, 3>), dim3(S, MS_PARTS), dim3(64), 0, s, a); \
      else if (mode == 1) hipLaunchKernelGGL((k_mixstate<DT_, 1>), dim3(S, MS_PARTS), dim3(64), 0, s, a); \
      else if (mode == 2) hipLaunchKernelGGL((k_mixstate<DT_, 2>), dim3(S, MS_PARTS), dim3(64), 0, s, a); \
      else hipLaunchKernelGGL((k_mixstate<DT_, 0>), dim3(S, MS_PARTS), dim3(64), 0, s, a); \
   } while (0)
   if (a.D == 39) MS_LAUNCH(39); else if (a.D == 26) MS_LAUNCH(26); else MS_LAUNCH(13);
#undef MS_LAUNCH
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

static int set_lds_limit(const void *fn, size_t lds)
{
   if (lds > 64 * 1024)      // dynamic LDS beyond 64 KiB must be allowed explicitly (160 KiB per CU on gfx950)
      HIPCHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
   return HTKAMD_OK;
}

int htkamd_launch_beta(const FbArgs &a, int blockDim, size_t lds, hipStream_t s)
{
   int rc = set_lds_limit((const void *)k_beta, lds);
   if (rc) return rc;
   hipLaunchKernelGGL(k_beta, dim3(a.nList), dim3(blockDim), lds, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_launch_alpha(const FbArgs &a, int blockDim, size_t lds, hipStream_t s)
{
   int rc = set_lds_limit((const void *)k_alpha, lds);
   if (rc) return rc;
   hipLaunchKernelGGL(k_alpha, dim3(a.nList), dim3(blockDim), lds, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// statistics of the surviving pairs: from the dense seed array (gamTotal seeds; the utterances off the left-to-right path) and / or from
// the hit list of k_stats_lr (a.hits: nHitsMax > 0), then the per-Gaussian reduction of the records both of them listed
int htkamd_launch_mixstats(const FbArgs &a_in, hipStream_t s, bool dense, bool listed, bool deferState)
{
   if (!dense && !listed) return HTKAMD_OK;
   FbArgs a = a_in;
   dense = dense && a.gamTotal > 0;
   // bucketed by state and nothing from the dense seed array: the list holds only what a full bucket turned away -- rarely anything -- and is
   // taken by ONE launch of the list kernel with direct atomics, which leaves at once when the counter of such pairs is zero (round 5: the
   // record path's five launches ran empty in every pass, 24 us of nothing), in front of the states so that what it adds is in place when a
   // range of states is called final
   const bool stateOnly = listed && !dense && htkamd_mixstate_applies(a);
   if (stateOnly) a.rec = nullptr;
   if (a.rec) HIPCHECK(hipMemsetAsync(a.recCtl, 0, sizeof(int) * (3 * ((size_t)a.G + 1) + 1), s));
#define MIX_LAUNCH(K, GS) \
   switch (a.D) { \
   case 39: hipLaunchKernelGGL((K<39, GS>), grid, block, 0, s, a); break; \
   case 26: hipLaunchKernelGGL((K<26, GS>), grid, block, 0, s, a); break; \
   case 13: hipLaunchKernelGGL((K<13, GS>), grid, block, 0, s, a); break; \
   default: hipLaunchKernelGGL((K<0, GS>), grid, block, 0, s, a); break; \
   }
#define MIX_LAUNCH_GS(K) \
   if (a.maxM <= 16) { MIX_LAUNCH(K, 16) } else if (a.maxM <= 32) { MIX_LAUNCH(K, 32) } else { MIX_LAUNCH(K, 64) }
   if (dense) {
      size_t waves = (a.gamTotal + 511) / 512;
      size_t blocks = (waves + 3) / 4;
      if (blocks > 8192) blocks = 8192;          // grid-stride beyond 32 waves per CU
      const dim3 grid((unsigned)blocks), block(256);
      MIX_LAUNCH_GS(k_mixstats)
   }
   if (listed) {
      // (bucketed by state where that applies: k_mixstate; the list then only holds what the buckets had no room for)
      const dim3 grid(stateOnly ? 512 : 4096), block(256);          // grid-stride over the blocks of the list (its length is on the device)
      if (stateOnly) {
         MIX_LAUNCH_GS(k_mixhits)
         HIPCHECK(hipGetLastError());
         if (!deferState) { const int rc = htkamd_launch_mixstate_range(a_in, 0, a_in.nTiedStates, s); if (rc) return rc; }
      } else {
         if (htkamd_mixstate_applies(a)) { const int rc = htkamd_launch_mixstate_range(a, 0, a.nTiedStates, s); if (rc) return rc; }
         MIX_LAUNCH_GS(k_mixhits)
      }
   }
#undef MIX_LAUNCH_GS
#undef MIX_LAUNCH
   HIPCHECK(hipGetLastError());
   if (a.rec) {
      const int nTile = (a.G + REC_TILE - 1) / REC_TILE;
      hipLaunchKernelGGL(k_rec_tilesum, dim3(nTile), dim3(256), 0, s, a.recCtl, a.G, a.recCtl + 1 + 3 * (a.G + 1));
      hipLaunchKernelGGL(k_rec_tilescan, dim3(nTile), dim3(256), 0, s, a.recCtl, a.G, a.recCtl + 1 + 3 * (a.G + 1));
      hipLaunchKernelGGL(k_rec_scatter, dim3(2048), dim3(256), 0, s, a);
      const int gb = (a.G + 3) / 4;
      hipLaunchKernelGGL(k_rec_reduce, dim3(gb < 16384 ? gb : 16384), dim3(256), 0, s, a);
      HIPCHECK(hipGetLastError());
   }
   return HTKAMD_OK;
}

// ---- several streams (HFB.c:1026-1066 Setotprob with S > 1, :1499-1602 UpMixParms' stream loop) -------------------------------
// The scoring kernels leave one row per (stream, chain state) in outpU; the state's log probability is their float sum in stream
// order (`sum += outprobj[s][0]`, :1057).  The reference then REPLACES every stream's value by "the sum of the other streams",
// sum - x_s in float (:1062-1064), which is what UpMixParms adds to a component's log posterior (:1611).  (Its second visit to a
// tied state at the same frame reads those replaced values back and halves their sum, :1059 -- equal to the first visit's value for
// S = 3 up to float rounding, and a different number for any other S: a defect this path does not reproduce; oracle/ does.)
__global__ __launch_bounds__(256) void k_combine_streams(FbArgs a)
{
   const UttDesc *up = a.utt + blockIdx.x;
   const size_t n = (size_t)up->T * up->nSlots;
   const float *src = a.outpU + up->outp0 * a.NSt;
   float *dst = a.outp + up->outp0;
   for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.y * blockDim.x) {
      float sum = 0.0f;
      for (int k = 0; k < a.NSt; k++) sum += src[(size_t)k * n + i];
      dst[i] = sum;
   }
}

int htkamd_launch_combine_streams(const FbArgs &a, hipStream_t s)
{
   if (a.nUtt <= 0) return HTKAMD_OK;
   hipLaunchKernelGGL(k_combine_streams, dim3((unsigned)a.nUtt, 16), dim3(256), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// UpMixParms for S > 1: the surviving (frame, chain state) pairs of the dense seed array, a wavefront per pair, lane = component for
// the posteriors and lane = dimension for the first- and second-order sums (only the dimensions of the component's stream).
// seed = initx (the state's log occupation without its own output probability) when some stream of the set has several components,
// else log alpha + log beta - pr.
// one surviving (frame, chain state) pair of a multi-stream set: every stream's components, posteriors and sums (by ONE wavefront)
__device__ void ms_pair(const FbArgs &a, const UttDesc *up, const int t0, const int slot, const double seed, const int lane)
{
   const int D = a.D, NSt = a.NSt;
   const double minF = (double)a.minFrwdP;
   const bool upMu = a.uFlags & HTKAMD_UPMEANS, upVa = a.uFlags & HTKAMD_UPVARS, upWt = a.uFlags & HTKAMD_UPMIXES;
   const int nSl = up->nSlots, T = up->T;
   const int e0 = a.slotState[up->slot0 + slot];
   const float *xrow = a.X + (size_t)(up->frame0 + t0) * D;
   const float oS = a.outp[up->outp0 + (size_t)slot * T + t0];
   for (int ks = 0; ks < NSt; ks++) {
      const int e = e0 + ks, c0 = a.stateCompOff[e], M = a.stateCompOff[e + 1] - c0;
      const float oK = a.outpU[up->outp0 * NSt + ((size_t)ks * nSl + slot) * T + t0];
      float sumS = oS;
      if (a.compatRevisit) {                                     // (the state's row may hold a second visit's value: the shared vectors keep the first visit's)
         sumS = 0.0f;
         for (int k2 = 0; k2 < NSt; k2++) sumS += a.outpU[up->outp0 * NSt + ((size_t)k2 * nSl + slot) * T + t0];
      }
      const float others = sumS - oK;                            // outprob[s][0] after Setotprob (HFB.c:1064)
      for (int mb = 0; mb < M; mb += 64) {
         const int m = mb + lane;
         bool pass = false;
         double Lr = 0.0;
         int g = 0;
         if (m < M) {
            const float wt = a.compLogWt[c0 + m];
            g = a.compGauss[c0 + m];
            if (wt > (float)LMINMIX) {
               double x;
               if (M == 1) x = (a.maxM == 1) ? seed : seed + (double)oS;        // !mmix: log alpha + log beta - pr (HFB.c:1584)
               else {
                  const float *P = a.gparam + (size_t)g * a.PS;
                  float sum = P[2 * D];
                  for (int k = 0; k < D; k++) {
                     const float xmm = xrow[k] - P[2 * k];
                     sum += xmm * xmm * P[2 * k + 1];
                  }
                  const float prob = -0.5f * sum;
                  x = seed + (double)wt;
                  x += (double)prob;
                  x += (double)others;                                            // "adjust for parallel streams" (HFB.c:1611)
               }
               if (-x < minF) { pass = true; Lr = exp(x); }
            }
         }
         double sumLr = pass ? Lr : 0.0;
#pragma unroll
         for (int o = 32; o > 0; o >>= 1) sumLr += __shfl_xor(sumLr, o);
         if (lane == 0 && sumLr != 0.0) atomicAdd(a.acc + a.lay.wtOcc + e, sumLr);
         if (pass) {
            if (upMu) atomicAdd(a.acc + a.lay.muOcc + g, Lr);
            if (upVa) atomicAdd(a.acc + a.lay.vaOcc + g, Lr);
            if (upWt) atomicAdd(a.acc + a.lay.wt + c0 + m, Lr);
         }
         unsigned long long pm = __ballot(pass);
         while (pm) {
            const int ml = __ffsll((long long)pm) - 1;
            pm &= pm - 1;
            const double L = __shfl(Lr, ml);
            const int gg = __shfl(g, ml);
            const float *mean = a.mean + (size_t)gg * D;
            for (int k = lane; k < D; k += 64) {
               if (a.dimStream[k] != ks) continue;
               const float z = xrow[k] - mean[k];
               if (upMu && upVa) {                    // HFB.c:1673-1678
                  const float zl = (float)((double)z * L);
                  atomicAdd(a.acc + a.lay.mu + (size_t)gg * D + k, (double)zl);
                  atomicAdd(a.acc + a.lay.va + (size_t)gg * D + k, (double)(z * zl));
               } else if (upMu) atomicAdd(a.acc + a.lay.mu + (size_t)gg * D + k, (double)z * L);
               else if (upVa) atomicAdd(a.acc + a.lay.va + (size_t)gg * D + k, (double)(z * z) * L);
            }
         }
      }
   }
}

__global__ __launch_bounds__(256) void k_mixstats_ms(FbArgs a)
{
   __shared__ unsigned short hitIdx[4][512];
   __shared__ double hitSeed[4][512];
   const int lane = threadIdx.x & 63;
   const size_t nWaves = ((size_t)gridDim.x * blockDim.x) >> 6;
   const size_t waveId = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   for (size_t base0 = waveId * 512; base0 < a.gamTotal; base0 += nWaves * 512) {
      volatile unsigned short *hIdx = hitIdx[threadIdx.x >> 6];
      volatile double *hSeed = hitSeed[threadIdx.x >> 6];
      int count = 0;
      for (int r = 0; r < 8; r++) {
         const size_t idx = base0 + (size_t)r * 64 + lane;
         const double v = (idx < a.gamTotal) ? a.gam[idx] : LZERO;
         const unsigned long long hm = __ballot(v > LSMALL);
         if (v > LSMALL) {
            const int pos = count + __popcll(hm & ((1ull << lane) - 1));
            hIdx[pos] = (unsigned short)(r * 64 + lane); hSeed[pos] = v;
         }
         count += __popcll(hm);
      }
      int u = a.gamChunkUtt[base0 >> 9];
      for (int i = 0; i < count; i++) {
         const size_t hidx = base0 + hIdx[i];
         const double seed = hSeed[i];
         while (u + 1 < a.nUtt && a.gamOffByUtt[u + 1] <= hidx) u++;
         const UttDesc *up = a.utt + u;
         if (a.status[u] != HTKAMD_UTT_OK || up->pad == 2) continue;      // (an utterance of the left-to-right path has no seeds here: its pairs are listed)
         const int nSl_ = up->nSlots;
         const size_t rel = hidx - up->gam0;
         ms_pair(a, up, (int)(rel / nSl_), (int)(rel % nSl_), seed, lane);
      }
   }
}

__device__ void tm_pair(const FbArgs &a, const UttDesc *up, const int t0, const int slot, const double seed, const int lane);
// The same from the LISTS of k_stats_lr (left-to-right chains, round 4): a wavefront per region, a pair after the other.  A record of
// these sets carries the pair's GLOBAL slot (FbArgs::hitSlots) -- the statistics want the chain state's own stream scores -- and the
// utterance is found from it (the slots of a batch ascend with the utterances).
template <bool TIED>
__global__ __launch_bounds__(256) void k_mixhits_streams(FbArgs a)
{
   const int lane = threadIdx.x & 63;
   const int nWaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
   const int waveId = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
   for (int b = waveId; b < a.nHitRegions; b += nWaves) {
      const int cnt = a.hitCtl[b];
      const MixHit *reg = a.hits + (size_t)b * a.hitRegionCap;
      int u = 0;
      for (int i = 0; i < cnt; i++) {
         const MixHit h = reg[i];
         if (i == 0 || h.st < a.utt[u].slot0 || h.st >= a.utt[u].slot0 + a.utt[u].nSlots) {      // (a region lies within one utterance: searched once)
            int lo = 0, hi = a.nUtt - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.utt[mid].slot0 <= h.st) lo = mid; else hi = mid - 1; }
            u = lo;
         }
         const UttDesc *up = a.utt + u;
         if (a.status[u] != HTKAMD_UTT_OK) continue;
         if (TIED) tm_pair(a, up, h.frame - up->frame0, h.st - up->slot0, h.seed, lane);
         else ms_pair(a, up, h.frame - up->frame0, h.st - up->slot0, h.seed, lane);
      }
   }
}

int htkamd_launch_mixhits_streams(const FbArgs &a, bool tied, hipStream_t s)
{
   if (a.nHitRegions <= 0) return HTKAMD_OK;
   if (tied) hipLaunchKernelGGL(k_mixhits_streams<true>, dim3(4096), dim3(256), 0, s, a);
   else hipLaunchKernelGGL(k_mixhits_streams<false>, dim3(4096), dim3(256), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_launch_mixstats_ms(const FbArgs &a, hipStream_t s)
{
   if (a.gamTotal == 0) return HTKAMD_OK;
   size_t waves = (a.gamTotal + 511) / 512;
   size_t blocks = (waves + 3) / 4;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_mixstats_ms, dim3((unsigned)blocks), dim3(256), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// ---- tied mixtures (hsKind TIEDHS) ------------------------------------------------------------------------------------------
// PrecomputeTMix (HModel.c:5308-5346, called with tmThresh = minFrwdP and topM = 0 from HFB.c:1011,1781): per frame and stream the log
// probability of every pool Gaussian (MOutP -> DOutP: the set keeps its variances), their maximum, and for those within tmThresh of
// it exp(p - max); the others are out (marked -1 here: the reference stops at the first of them in its sorted list).
// One workgroup per (frame, stream), thread = pool Gaussian.
__global__ __launch_bounds__(256) void k_tm_pool(FbArgs a)
{
   __shared__ float red[256];
   const int row = blockIdx.x, ks = blockIdx.y, D = a.D;
   const int p0 = a.tmPoolOff[ks], M = a.tmPoolOff[ks + 1] - p0;
   const int c0 = a.stateCompOff[ks];                         // the pool of stream ks, as state 0 lists it
   const float *x = a.X + (size_t)row * D;
   float mx = (float)LZERO;
   for (int m = threadIdx.x; m < M; m += blockDim.x) {
      const int g = a.compGauss[c0 + m];
      const float *mean = a.mean + (size_t)g * D, *var = a.var + (size_t)g * D;
      float sum = a.gparam[(size_t)g * a.PS + 2 * D];           // gConst
      for (int k = 0; k < D; k++) {
         if (a.dimStream && a.dimStream[k] != ks) continue;
         const float xmm = x[k] - mean[k];
         sum += xmm * xmm / var[k];
      }
      const float p = -0.5f * sum;
      a.tmE[(size_t)row * a.tmPool + p0 + m] = p;
      mx = p > mx ? p : mx;
   }
   red[threadIdx.x] = mx;
   __syncthreads();
   for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
   const float maxP = red[0];
   const float minP = maxP - a.minFrwdP;
   if (threadIdx.x == 0) a.tmMaxP[(size_t)row * a.NSt + ks] = maxP;
   for (int m = threadIdx.x; m < M; m += blockDim.x) {
      float *e = a.tmE + (size_t)row * a.tmPool + p0 + m;
      const float p = *e;
      if (p < minP) *e = -1.0f;
      else { const float dp = p - maxP; *e = ((double)dp < MINEARG) ? 0.0f : (float)exp((double)dp); }
   }
}

// SOutP, TIEDHS (HModel.c:5555-5566): sum over the kept pool entries of (scaled probability x weight), float products summed in
// double, log + maximum.  (The reference adds them in the order of its sorted list; the order here is the pool's: the double sum can
// differ in its last bit.)  One workgroup per scoring task, thread = frame of the tile.
__global__ __launch_bounds__(SCORE_TILE_FRAMES) void k_tm_state(FbArgs a)
{
   for (int task = blockIdx.x; task < a.tmNTasks; task += gridDim.x) {
      const ScoreTask tk = a.tmTasks[task];
      const int f = threadIdx.x;
      if (f >= tk.nFrames) continue;
      const size_t row = (size_t)tk.frame0 + f;
      for (int k = 0; k < tk.nSlots; k++) {
         const int e0 = a.tmSlotState[tk.slot0 + k];
         float total = 0.0f, xs = 0.0f;
         for (int kk = 0; kk < (a.tmCombine ? a.NSt : 1); kk++) {
            const int e = e0 + kk, ks = e % a.NSt;
            const int c0 = a.stateCompOff[e], p0 = a.tmPoolOff[ks], M = a.tmPoolOff[ks + 1] - p0;
            const float *E = a.tmE + row * a.tmPool + p0, *w = a.compWeight + c0;
            double sum = 0.0;
            for (int m = 0; m < M; m++) {
               const float ev = E[m];
               if (ev >= 0.0f) sum += (double)(ev * w[m]);
            }
            xs = (sum >= MINLARG) ? (float)(log(sum) + (double)a.tmMaxP[row * a.NSt + ks]) : (float)LZERO;
            if (a.tmCombine && a.NSt > 1) { const float wx = a.streamWt[e] * xs; total = total + wx; }      // cPOutP: float product, float sum
         }
         a.tmOut[tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + f] = (a.tmCombine && a.NSt > 1) ? total : xs;
      }
   }
}

int htkamd_launch_tm_score(const FbArgs &a, hipStream_t s)
{
   if (a.totalFrames <= 0 || a.tmNTasks <= 0) return HTKAMD_OK;
   hipLaunchKernelGGL(k_tm_pool, dim3((unsigned)a.totalFrames, (unsigned)a.NSt), dim3(256), 0, s, a);
   hipLaunchKernelGGL(k_tm_state, dim3((unsigned)(a.tmNTasks < 65535 ? a.tmNTasks : 65535)), dim3(SCORE_TILE_FRAMES), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_tm_score_block(const htkamd_model *m, const ScoreArgs &sa, int nRows, float tmBeam, hipStream_t s)
{
   if (nRows <= 0 || sa.nTasks <= 0) return HTKAMD_OK;
   FbArgs a;
   memset(&a, 0, sizeof(a));
   float *tmE = nullptr, *tmMaxP = nullptr;
   HIPCHECK(hipMalloc((void **)&tmE, sizeof(float) * ((size_t)nRows * m->tmPool + 16)));
   hipError_t e = hipMalloc((void **)&tmMaxP, sizeof(float) * ((size_t)nRows * m->NSt + 16));
   if (e != hipSuccess) { (void)hipFree(tmE); htkamd_set_error("tm_score_block: hipMalloc: %s", hipGetErrorString(e)); return HTKAMD_ENOMEM; }
   a.X = sa.X; a.D = m->D; a.PS = m->PS; a.gparam = m->d_gparam; a.mean = m->d_mean; a.var = m->d_var; a.compGauss = m->d_compGauss; a.stateCompOff = m->d_stateCompOff;
   a.dimStream = m->d_dimStream; a.NSt = m->NSt; a.tmE = tmE; a.tmMaxP = tmMaxP; a.tmPoolOff = m->d_tmPoolOff; a.tmPool = m->tmPool; a.totalFrames = nRows;
   a.minFrwdP = tmBeam;                                   // PrecomputeTMix's tmThresh: HVite -c (HVite.c:115,255), HRec.c:1987
   a.tmTasks = sa.tasks; a.tmNTasks = sa.nTasks; a.tmSlotState = sa.slotState; a.tmOut = sa.out; a.compWeight = m->d_compWeight;
   a.tmCombine = 1; a.streamWt = m->d_streamWt;
   int rc = htkamd_launch_tm_score(a, s);
   hipError_t e2 = hipStreamSynchronize(s);
   (void)hipFree(tmE); (void)hipFree(tmMaxP);
   if (!rc && e2 != hipSuccess) { htkamd_set_error("tm_score_block: %s", hipGetErrorString(e2)); rc = HTKAMD_EHIP; }
   return rc;
}

// UpMixParms, TIEDHS (HFB.c:1503-1507, 1559-1563, 1590-1612): the kept pool entries of the frame; a component's log probability is
// log(scaled probability) + maximum, as the reference recovers it from the table PrecomputeTMix left
// one surviving (frame, chain state) pair of a tied-mixture set: the kept pool entries of every stream (by ONE wavefront)
__device__ void tm_pair(const FbArgs &a, const UttDesc *up, const int t0, const int slot, const double seed, const int lane)
{
   const int D = a.D, NSt = a.NSt;
   const double minF = (double)a.minFrwdP;
   const bool upMu = a.uFlags & HTKAMD_UPMEANS, upVa = a.uFlags & HTKAMD_UPVARS, upWt = a.uFlags & HTKAMD_UPMIXES;
   const int nSl = up->nSlots, T = up->T;
   const int e0 = a.slotState[up->slot0 + slot];
   const size_t row = (size_t)up->frame0 + t0;
   const float *xrow = a.X + row * D;
   const float oS = a.outp[up->outp0 + (size_t)slot * T + t0];
   for (int ks = 0; ks < NSt; ks++) {
      const int e = e0 + ks, c0 = a.stateCompOff[e], p0 = a.tmPoolOff[ks], M = a.tmPoolOff[ks + 1] - p0;
      const float maxP = a.tmMaxP[row * NSt + ks];
      float others = 0.0f;
      if (NSt > 1) others = oS - a.outpU[up->outp0 * NSt + ((size_t)ks * nSl + slot) * T + t0];
      for (int mb = 0; mb < M; mb += 64) {
         const int m = mb + lane;
         bool pass = false;
         double Lr = 0.0;
         int g = 0;
         if (m < M) {
            const float ev = a.tmE[row * a.tmPool + p0 + m];
            const float wt = a.compLogWt[c0 + m];
            g = a.compGauss[c0 + m];
            if (ev >= 0.0f && wt > (float)LMINMIX) {
               const float prob = ((double)ev >= MINLARG) ? (float)(log((double)ev) + (double)maxP) : (float)LZERO;
               double x = seed + (double)wt;
               x += (double)prob;
               if (NSt > 1) x += (double)others;
               if (-x < minF) { pass = true; Lr = exp(x); }
            }
         }
         double sumLr = pass ? Lr : 0.0;
#pragma unroll
         for (int o = 32; o > 0; o >>= 1) sumLr += __shfl_xor(sumLr, o);
         if (lane == 0 && sumLr != 0.0) atomicAdd(a.acc + a.lay.wtOcc + e, sumLr);
         if (pass) {
            if (upMu) atomicAdd(a.acc + a.lay.muOcc + g, Lr);
            if (upVa) atomicAdd(a.acc + a.lay.vaOcc + g, Lr);
            if (upWt) atomicAdd(a.acc + a.lay.wt + c0 + m, Lr);
         }
         unsigned long long pm = __ballot(pass);
         while (pm) {
            const int ml = __ffsll((long long)pm) - 1;
            pm &= pm - 1;
            const double L = __shfl(Lr, ml);
            const int gg = __shfl(g, ml);
            const float *mean = a.mean + (size_t)gg * D;
            for (int k = lane; k < D; k += 64) {
               if (a.dimStream && a.dimStream[k] != ks) continue;
               const float z = xrow[k] - mean[k];
               if (upMu && upVa) {
                  const float zl = (float)((double)z * L);
                  atomicAdd(a.acc + a.lay.mu + (size_t)gg * D + k, (double)zl);
                  atomicAdd(a.acc + a.lay.va + (size_t)gg * D + k, (double)(z * zl));
               } else if (upMu) atomicAdd(a.acc + a.lay.mu + (size_t)gg * D + k, (double)z * L);
               else if (upVa) atomicAdd(a.acc + a.lay.va + (size_t)gg * D + k, (double)(z * z) * L);
            }
         }
      }
   }
}

__global__ __launch_bounds__(256) void k_mixstats_tm(FbArgs a)
{
   __shared__ unsigned short hitIdx[4][512];
   __shared__ double hitSeed[4][512];
   const int lane = threadIdx.x & 63;
   const size_t nWaves = ((size_t)gridDim.x * blockDim.x) >> 6;
   const size_t waveId = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   for (size_t base0 = waveId * 512; base0 < a.gamTotal; base0 += nWaves * 512) {
      volatile unsigned short *hIdx = hitIdx[threadIdx.x >> 6];
      volatile double *hSeed = hitSeed[threadIdx.x >> 6];
      int count = 0;
      for (int r = 0; r < 8; r++) {
         const size_t idx = base0 + (size_t)r * 64 + lane;
         const double v = (idx < a.gamTotal) ? a.gam[idx] : LZERO;
         const unsigned long long hm = __ballot(v > LSMALL);
         if (v > LSMALL) {
            const int pos = count + __popcll(hm & ((1ull << lane) - 1));
            hIdx[pos] = (unsigned short)(r * 64 + lane); hSeed[pos] = v;
         }
         count += __popcll(hm);
      }
      int u = a.gamChunkUtt[base0 >> 9];
      for (int i = 0; i < count; i++) {
         const size_t hidx = base0 + hIdx[i];
         const double seed = hSeed[i];
         while (u + 1 < a.nUtt && a.gamOffByUtt[u + 1] <= hidx) u++;
         const UttDesc *up = a.utt + u;
         if (a.status[u] != HTKAMD_UTT_OK || up->pad == 2) continue;      // (an utterance of the left-to-right path has no seeds here: its pairs are listed)
         const int nSl_ = up->nSlots;
         const size_t rel = hidx - up->gam0;
         tm_pair(a, up, (int)(rel / nSl_), (int)(rel % nSl_), seed, lane);
      }
   }
}

int htkamd_launch_mixstats_tm(const FbArgs &a, hipStream_t s)
{
   if (a.gamTotal == 0) return HTKAMD_OK;
   size_t waves = (a.gamTotal + 511) / 512;
   size_t blocks = (waves + 3) / 4;
   if (blocks > 8192) blocks = 8192;
   hipLaunchKernelGGL(k_mixstats_tm, dim3((unsigned)blocks), dim3(256), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}
