// fb_wave.hip -- K2w / K3w: beta and alpha passes with ONE WAVEFRONT PER UTTERANCE.
//
// Same reference semantics as fb_kernels.hip (SetBeta, StepAlpha, InitAlpha, MaxModelProb, SetOcct,
// UpTranParms, UpMixParms seeds -- HFB.c, S==1) for utterances whose chain has at most 64*W models (W = 1, 2, 4 or 8
// wavefronts per utterance) of at most MAXN states.  This is the fast path; fb_kernels.hip (one workgroup per
// utterance, cells in LDS) stays as the general one.  With W > 1 the wavefronts of an utterance exchange the few
// values that cross a 64-model boundary (entry/exit values of the neighbouring models, beam ballots, the column
// maximum) through LDS with one workgroup barrier per exchange (Grp<W> below); W = 1 compiles to plain wave
// shuffles and ballots with no barrier at all.
//
// MI355X mapping.  The recursions are a T-step dependent chain with only ~Q-way parallelism per step, so what
// limits them is the latency of one step, not throughput.  Here lane q of a 64-wide wavefront owns model q of
// the utterance and keeps everything that belongs to the model in registers with compile-time indexing: its
// transition matrix, the alpha/beta values of its N states for the current and previous frame, the output
// probabilities of its emitting states, and its transition counters.  A model talks to its neighbours only
// through the exit/entry state values (and those of the model after next, for tee models), which are passed
// with wave shuffles; beam decisions are ballots and bit scans on the 64-bit lane mask.  A time step therefore
// needs NO barrier and NO LDS traffic except the LAdd table, which the 4 wavefronts (= 4 utterances) of a
// workgroup share.  Everything that comes from HBM (beta column, output probabilities, beams of frame t+2)
// is requested two frames ahead and rotates through registers, so no step waits on memory; beta and the
// mixture seeds go out as per-lane contiguous runs (N doubles / N-2 doubles), i.e. coalesced across the wave.
#include <hip/hip_runtime.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "ladd.h"
#include "wavegrp.h"

#define UPB(W) ((W) == 1 ? 4 : 1)   // utterances per workgroup: four single-wave utterances, or one multi-wave utterance
#define EXPFLOOR (-100.0)     // see fb_kernels.hip

#define ladd(x, y) ladd_sel<FAST>((x), (y), mle, ltab)
#define EXPT(x) exp_sel<FAST>((x), etab)

template <int MAXN> struct ModelRegs {
   int N, mc0, ms0;
   float tp[MAXN][MAXN];      // tp[i-1][j-1] = log a_ij
   float aN[MAXN];            // aN[i-1] = log a_iN (exit transitions), aN[0] = a_1N (tee)
};

template <int MAXN>
__device__ __forceinline__ void load_model(ModelRegs<MAXN> &m, const FbArgs &a, const UttDesc &ud, int q, bool valid)
{
   m.N = 0; m.mc0 = 0; m.ms0 = 0;
#pragma unroll
   for (int i = 0; i < MAXN; i++) {
      m.aN[i] = (float)LZERO;
#pragma unroll
      for (int j = 0; j < MAXN; j++) m.tp[i][j] = (float)LZERO;
   }
   if (valid) {
      const int mi = ud.q0 + q - 1;
      m.N = a.mN[mi]; m.mc0 = a.mCell0[mi]; m.ms0 = a.mSlot0[mi];
      const float *tp = a.transP + a.mTp[mi];
#pragma unroll
      for (int i = 0; i < MAXN; i++)
#pragma unroll
         for (int j = 0; j < MAXN; j++)
            if (i < m.N && j < m.N) m.tp[i][j] = tp[i * m.N + j];
#pragma unroll
      for (int i = 0; i < MAXN; i++)
         if (i < m.N) m.aN[i] = tp[i * m.N + (m.N - 1)];
   }
}


// Output-probability rows are time-contiguous per state (outp[slot][t]), so a lane fetches FOUR frames of one of its states with a
// single 16-byte load.  A block is requested one block (4 frames) before it is needed, waits in registers, and is written to a
// wave-private LDS slot when the recursion reaches it; each frame then reads its scores from LDS.  No register is copied between
// request and use, so no step waits for a load younger than four frames (a per-frame 4-byte gather with one frame of look-ahead
// left the wave waiting on HBM latency every step).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NE> struct ObsStage {
   float *lds;                 // this wave's [2][NE][64][4] floats
   f4u R[NE];                  // block in flight
   const float *row[NE];       // the lane's score rows (NULL: no such state)
   int lane;
   __device__ __forceinline__ void load(int blk)
   {
#pragma unroll
      for (int j = 0; j < NE; j++) if (row[j]) R[j] = *(const f4u *)(row[j] + 4 * blk);
   }
   __device__ __forceinline__ void park(int blk)
   {
#pragma unroll
      for (int j = 0; j < NE; j++) *(f4u *)(lds + ((((blk & 1) * NE + j) * 64 + lane) << 2)) = R[j];
   }
   // scores of frame index f (0-based) into ob[2..]: the frame's block must have been parked
   __device__ __forceinline__ void get(int f, float *ob) const
   {
#pragma unroll
      for (int j = 0; j < NE; j++) if (row[j]) ob[j + 2] = lds[(((((f >> 2) & 1) * NE + j) * 64 + lane) << 2) + (f & 3)];
   }
};

// beta of the wave path: betaW[betaW0 + ((t-1)*MAXN + i-1)*64*W + group lane] -- one state of all models of a frame is one contiguous run,
// so the wave's store (beta pass) and load (alpha pass) of a state are coalesced
#define BETA_W(t, i) (gbeta[((size_t)((t) - 1) * MAXN + ((i) - 1)) * (64 * W)])

// ------------------------------------------------------------------------------------ K2w: beta
template <int MAXN, int W, bool FAST>
__global__ __launch_bounds__(64 * W * UPB(W)) void k_beta_w(FbArgs a)
{
   __shared__ double ltab[FAST ? 1 : LADD_TAB_DOUBLES];
   __shared__ float stage[W * UPB(W)][2 * (MAXN - 2) * 64 * 4];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   if constexpr (!FAST) ladd_table_to_lds(ltab, a.laddTab);
   __syncthreads();
   const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;           // wave in block
   const int li = blockIdx.x * UPB(W) + wib / W;                        // the block's utterance(s): all waves of a group leave together
   if (li >= a.nList) return;
   const int u = a.uttList[li];
   Grp<W> g; g.x = gx; g.a1 = ga1; g.wave = wib % W; g.lane = lane; g.ph = 0;
   const int gl = 64 * g.wave + lane;
   const UttDesc ud = a.utt[u];
   if (ud.status != HTKAMD_UTT_OK) {
      if (gl == 0) { a.status[u] = ud.status; a.pr[u] = LZERO; }
      return;
   }
   const int T = ud.T, Q = ud.Q;
   const int q = gl + 1;
   const bool valid = q <= Q;
   ModelRegs<MAXN> m;
   load_model<MAXN>(m, a, ud, q, valid);
   const int N = m.N;
   const float a1N = m.aN[0];
   if constexpr (W > 1) ga1[gl] = a1N;                   // read after the barrier of the first exchange below
   // neighbours' constants
   const float a1N_n1 = (float)g.down1((double)a1N);     // a_1N of model q+1 (LZERO-ish garbage beyond Q is masked below)
   const int dm = valid ? a.mDms[ud.q0 + q - 1] : 1;
   const MaskW<W> teeMask = g.ballot(valid && dm == 0);  // bit (q-1) set when model q is a tee model

   const short *tLo = a.taperLo + ud.frame0 - 1, *tHi = a.taperHi + ud.frame0 - 1;   // 1-based t
   short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
   ObsStage<MAXN - 2> st;
   st.lds = stage[wib]; st.lane = lane;
#pragma unroll
   for (int j = 0; j < MAXN - 2; j++) {
      st.row[j] = (valid && j + 2 < N) ? a.outp + ud.outp0 + (size_t)(m.ms0 + j) * T : nullptr;   // rows of this model's emitting states
      st.R[j] = (f4u)(0.f);
   }
   double *gbeta = a.betaW + ud.betaW0 + gl;
   const double mle = a.minLogExp;
   const bool pruning = a.pruneInit < HTKAMD_NOPRUNE;

   double thresh = a.pruneInit, pr = LZERO;
   int ok = 0;
   for (;;) {                                            // StepBack retry loop (HFB.c:1332-1361)
      int fail = 0;
      double bC[MAXN + 1], bP[MAXN + 1];                 // bC[i] = beta_i(t) of this model, i = 1..N
      float obT[MAXN], ob1[MAXN], obP[MAXN];             // b_j(t), b_j(t+1), prefetch of b_j(t-1); j = 2..N-1
#pragma unroll
      for (int i = 0; i <= MAXN; i++) { bC[i] = LZERO; bP[i] = LZERO; }
#pragma unroll
      for (int j = 0; j < MAXN; j++) { obT[j] = 0.f; ob1[j] = 0.f; obP[j] = 0.f; }
      // ---- t = T (HFB.c:1175-1198)
      int endq = tLo[T];
      {  // scores: the last block straight into LDS, the one before it in flight
         const int bl = (T - 1) >> 2;
         st.load(bl); st.park(bl);
         if (bl >= 1) st.load(bl - 1);
         st.get(T - 1, obT);
         if (T >= 2) {
            if (((T - 2) & 3) == 3) { st.park((T - 2) >> 2); if (((T - 2) >> 2) >= 1) st.load(((T - 2) >> 2) - 1); }
            st.get(T - 2, obP);
         }
      }
      {
         // exit chain: e(Q) = 0, e(q) = e(q+1) + a_1N(q+1)
         double e = 0.0, mine = 0.0;
         for (int k = Q; k >= endq; k--) {
            if (k < Q) e = e + (double)g.a1N_at(a1N, k);   // lane k holds model k+1
            if (q == k) mine = e;
         }
         if (valid && q >= endq) {
#pragma unroll
            for (int i = 1; i <= MAXN; i++) if (i == N) bC[i] = mine;
#pragma unroll
            for (int i = 2; i < MAXN; i++) if (i < N) bC[i] = (double)m.aN[i - 1] + mine;
            double x = LZERO;
#pragma unroll
            for (int j = 2; j < MAXN; j++)
               if (j < N) {
                  const double aa = m.tp[0][j - 1], y = bC[j];
                  if (aa > LSMALL && y > LSMALL) x = ladd(x, aa + (double)obT[j] + y);
               }
            bC[1] = x;
#pragma unroll
            for (int i = 1; i <= MAXN; i++) if (i <= N) BETA_W(T, i) = bC[i];
         }
      }
      if (gl == 0) { gLo[T] = (short)endq; gHi[T] = (short)Q; }
      int qHiN = Q, qLoN = endq, lastEnd = endq;
      int nxtLo = (T >= 2) ? tLo[T - 1] : 1, nxtHi = (T >= 2) ? tHi[T - 1] : 1;
      bool stPrev = false, stIn = false; int tPrev = 0, loPrev = 1, hiPrev = 1;

      // ---- t = T-1 .. 1 (HFB.c:1205-1277)
      for (int t = T - 1; t >= 1; t--) {
         const int taperLoT = nxtLo, taperHiT = nxtHi;
         if (t >= 2) { nxtLo = tLo[t - 1]; nxtHi = tHi[t - 1]; }
         // a new block of scores is parked when the recursion reaches it, and the next one requested; this is the only place the
         // wave waits for memory: the block's loads are four frames old, and the stores below are issued after it
         if (t >= 2 && ((t - 2) & 3) == 3) { st.park((t - 2) >> 2); if (((t - 2) >> 2) >= 1) st.load(((t - 2) >> 2) - 1); }
         // rotate: previous column, output probabilities
#pragma unroll
         for (int i = 0; i <= MAXN; i++) bP[i] = bC[i];
         // the column finished in the previous iteration goes out now (vmcnt counts stores on gfx9: a store issued at the end of a
         // frame would be waited for at the top of the next one)
         if (stPrev) {
            if (stIn) {
#pragma unroll
               for (int i = 1; i <= MAXN; i++) if (i <= N) BETA_W(tPrev, i) = bP[i];
            }
            if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; }
            stPrev = false;
         }
#pragma unroll
         for (int j = 0; j < MAXN; j++) { ob1[j] = obT[j]; obT[j] = obP[j]; }
         if (t >= 2) st.get(t - 2, obP);
         const int startq = qHiN;
         endq = (qLoN == 1) ? 1 : ((taperLoT >= qLoN) ? taperLoT : qLoN - 1);
         while (endq > 1 && teeMask.bit(endq - 2)) endq--;
         // neighbours' entry values of column t+1 (shuffles are executed by every lane)
         double e1, e2;
         g.down2(bP[1], e1, e2);
         const bool inRange = valid && q >= endq && q <= startq;
         double lMax = LZERO;
         if (inRange) {
            const bool p1 = (q < Q) && (q + 1 >= qLoN) && (q + 1 <= qHiN);
            double ex = p1 ? e1 : LZERO;
            if (q < startq && a1N_n1 > (float)LSMALL) {  // next model is a tee model: its same-frame exit value
               const bool p2 = (q + 1 < Q) && (q + 2 >= qLoN) && (q + 2 <= qHiN);
               const double ex1 = p2 ? e2 : LZERO;
               ex = ladd(ex, ex1 + (double)a1N_n1);
            }
            const bool wasIn = q >= qLoN && q <= qHiN;
#pragma unroll
            for (int i = 1; i <= MAXN; i++) if (i == N) bC[i] = ex;
#pragma unroll
            for (int i = MAXN - 1; i >= 2; i--)
               if (i < N) {
                  double x = (double)m.aN[i - 1] + ex;
                  if (wasIn) {
#pragma unroll
                     for (int j = 2; j < MAXN; j++)
                        if (j < N) {
                           const double aa = m.tp[i - 1][j - 1], y = bP[j];
                           if (aa > LSMALL && y > LSMALL) x = ladd(x, aa + (double)ob1[j] + y);
                        }
                  }
                  bC[i] = x;
               }
            double x = LZERO;
#pragma unroll
            for (int j = 2; j < MAXN; j++)
               if (j < N) {
                  const double aa = m.tp[0][j - 1], y = bC[j];
                  if (y > lMax) lMax = y;
                  if (aa > LSMALL && y > LSMALL) x = ladd(x, aa + (double)obT[j] + y);
               }
            bC[1] = x;
         }
         int newHi, newLo;
         if (!pruning) {                                 // only the taper acts (HFB.c:1259-1264)
            newHi = (taperHiT < startq) ? taperHiT : startq;
            newLo = endq;
         } else {                                        // beam pruning (HFB.c:1254-1272) on the lane mask
            const double gmax = g.maxall(inRange ? lMax : LZERO);
            const MaskW<W> keep = g.ballot(inRange && !(gmax - lMax > thresh));
            int s = (keep & MaskW<W>::range(0, startq - 1)).highest() + 1;       // model numbers are lane+1
            if (s >= 1 && taperHiT < s) s = taperHiT;
            if (s < 1) { fail = 1; newHi = newLo = 1; }
            else if (keep.bit(endq - 1)) { newHi = s; newLo = endq; }             // the bottom model survives: no test against s (the
            else {                                                                 // taper may have pulled s below it, HFB.c:1259-1272)
               const int e = (keep & MaskW<W>::range(endq, s - 1)).lowest() + 1;   // raise endq till thresh reached; passing s fails
               if (e < 1) { fail = 1; newHi = newLo = 1; }
               else { newHi = s; newLo = e; }
            }
         }
         if (fail) break;
         stPrev = true; stIn = inRange; tPrev = t; loPrev = newLo; hiPrev = newHi;
         qHiN = newHi; qLoN = newLo; lastEnd = endq;
      }
      if (!fail && stPrev) {                             // the last column (t = 1)
         if (stIn) {
#pragma unroll
            for (int i = 1; i <= MAXN; i++) if (i <= N) BETA_W(tPrev, i) = bC[i];
         }
         if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; }
      }
      if (!fail) {
         pr = g.bcast(bC[1], lastEnd - 1);               // utt->pr = bqt[1] of the last model processed
         if (pr > LSMALL) { ok = 1; break; }
      }
      thresh += a.pruneInc;
      if (thresh > a.pruneLim || a.pruneInc == 0.0) break;
   }
   if (gl == 0) {
      a.pr[u] = ok ? pr : LZERO;
      a.status[u] = ok ? HTKAMD_UTT_OK : HTKAMD_UTT_SKIPPED;
   }
}

// ------------------------------------------------------------------------------------ K3w: alpha + stats
template <int MAXN, int W, bool FAST>
__global__ __launch_bounds__(64 * W * UPB(W)) void k_alpha_w(FbArgs a)
{
   __shared__ double ltab[FAST ? 1 : LADD_TAB_DOUBLES];
   __shared__ double etab[FAST ? 1 : EXP_TAB_N];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   if constexpr (!FAST) { ladd_table_to_lds(ltab, a.laddTab); exp_table_to_lds(etab); }
   __syncthreads();
   const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
   const int li = blockIdx.x * UPB(W) + wib / W;
   if (li >= a.nList) return;
   const int u = a.uttList[li];
   Grp<W> g; g.x = gx; g.a1 = ga1; g.wave = wib % W; g.lane = lane; g.ph = 0;
   const int gl = 64 * g.wave + lane;
   const UttDesc ud = a.utt[u];
   if (a.status[u] != HTKAMD_UTT_OK) {                   // skipped in the beta pass (or pre-check)
      if (gl == 0) atomicAdd(a.acc + a.lay.nUttSkipped, 1.0);
      return;
   }
   const int T = ud.T, Q = ud.Q, nC = ud.nCells, nSlots = ud.nSlots;
   const int q = gl + 1;
   const bool valid = q <= Q;
   ModelRegs<MAXN> m;
   load_model<MAXN>(m, a, ud, q, valid);
   const int N = m.N;
   const float a1N = m.aN[0];
   if constexpr (W > 1) ga1[gl] = a1N;
   const float a1N_p1 = (float)g.up1((double)a1N);       // a_1N of model q-1
   const int dm = valid ? a.mDms[ud.q0 + q - 1] : 1;
   const MaskW<W> teeMask = g.ballot(valid && dm == 0);
   const int cHmm = valid ? a.mHmm[ud.q0 + q - 1] : 0, cTrans = valid ? a.mTrans[ud.q0 + q - 1] : 0;
   int cM[MAXN];                                          // mixture count of emitting state j
#pragma unroll
   for (int j = 0; j < MAXN; j++) cM[j] = 0;
   if (valid) {
#pragma unroll
      for (int j = 2; j < MAXN; j++)
         if (j < N) {
            const int s = a.slotState[ud.slot0 + m.ms0 + j - 2];
            cM[j] = a.stateCompOff[s + 1] - a.stateCompOff[s];
         }
   }
   const short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;    // final beta beam, 1-based t
   short *gaLo = a.aLo + ud.frame0 - 1, *gaHi = a.aHi + ud.frame0 - 1;
   const float *orow = a.outp + ud.outp0 + (size_t)m.ms0 * T;
   const double *gbeta = a.betaW + ud.betaW0 + gl;
   double *gam = a.gam + ud.gam0 + m.ms0;
   const double mle = a.minLogExp, pr = a.pr[u];
   const double minF = (double)a.minFrwdP;
   const bool wantMix = (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES)) != 0;
   const bool wantTrans = (a.uFlags & HTKAMD_UPTRANS) != 0;

   double aC[MAXN + 1], aP[MAXN + 1];                    // alpha_i(t), alpha_i(t-1)
   double bT[MAXN + 1], bT1[MAXN + 1], bT2[MAXN + 1];    // beta_i(t), (t+1), (t+2: in flight)
   float oT[MAXN], oT1[MAXN], oT2[MAXN];                 // b_j(t), (t+1), (t+2: in flight)
   double xpre[MAXN];                                    // log sum_i alpha_i(t-1) a_ij (+ entry) before b_j(t)
   double ta[MAXN][MAXN + 1];                            // transition counts of this model: ta[i][j], i = 1..N-1, j = 2..N
   double occAcc[MAXN];
#pragma unroll
   for (int i = 0; i <= MAXN; i++) { aC[i] = LZERO; aP[i] = LZERO; bT[i] = LZERO; bT1[i] = LZERO; bT2[i] = LZERO; }
#pragma unroll
   for (int j = 0; j < MAXN; j++) {
      oT[j] = 0.f; oT1[j] = 0.f; oT2[j] = 0.f; xpre[j] = LZERO; occAcc[j] = 0.0;
#pragma unroll
      for (int k = 0; k <= MAXN; k++) ta[j][k] = 0.0;
   }
   // beams of frames t-1, t, t+1 and (in flight) t+2
   int lo0 = 1, hi0 = 0, lo1 = gLo[1], hi1 = gHi[1], lo2 = (T >= 2) ? gLo[2] : 1, hi2 = (T >= 2) ? gHi[2] : 0, lo3 = 1, hi3 = 0;
   if (valid) {
#pragma unroll
      for (int i = 1; i <= MAXN; i++)
         if (i <= N) { bT[i] = BETA_W(1, i); if (T >= 2) bT1[i] = BETA_W(2, i); }
#pragma unroll
      for (int j = 2; j < MAXN; j++)
         if (j < N) { oT[j] = orow[(size_t)(j - 2) * T]; if (T >= 2) oT1[j] = orow[(size_t)(j - 2) * T + 1]; }
   }
   int sq = 1, eq = hi1, err = 0;
   double mmpA = LZERO;                                  // MaxModelProb(q, t, minq = q) of the column just finished
   double exitSumLast = LZERO;                           // alpha_N + beta_N of this model in that column

   // ---- t = 1: InitAlpha (HFB.c:616-651)
   {
      double a1 = 0.0, mine = 0.0;
      for (int k = 1; k <= eq; k++) {
         if (k > 1) a1 = a1 + (double)g.a1N_at(a1N, k - 2);            // lane k-2 holds model k-1
         if (q == k) mine = a1;
      }
      if (valid && q <= eq) {
         aC[1] = mine;
#pragma unroll
         for (int j = 2; j < MAXN; j++)
            if (j < N) {
               const double aa = m.tp[0][j - 1];
               xpre[j] = aC[1] + aa;
               aC[j] = (aa > LSMALL) ? xpre[j] + (double)oT[j] : LZERO;
            }
         double x = LZERO;
#pragma unroll
         for (int i = 2; i < MAXN; i++)
            if (i < N) {
               const double aa = m.aN[i - 1];
               if (aa > LSMALL) x = ladd(x, aC[i] + aa);
            }
#pragma unroll
         for (int i = 1; i <= MAXN; i++) if (i == N) aC[i] = x;
      }
   }

   for (int t = 1; t <= T; t++) {
      // request column t+2 (lands in bT2/oT2, rotated in at the end of the step)
      if (t + 2 <= T) {
         lo3 = gLo[t + 2]; hi3 = gHi[t + 2];
         if (valid) {
#pragma unroll
            for (int i = 1; i <= MAXN; i++) if (i <= N) bT2[i] = BETA_W(t + 2, i);
#pragma unroll
            for (int j = 2; j < MAXN; j++) if (j < N) oT2[j] = orow[(size_t)(j - 2) * T + (t + 1)];
         }
      }
      if (t > 1) {
         // ---- alpha beam (HFB.c:699-722) from MaxModelProb of column t-1, as lane masks
         // lo0/hi0 = beta beam of t-1, lo1/hi1 = beta beam of t
         const MaskW<W> pruneA = g.ballot(valid && (pr - mmpA > minF));
         int s = (~pruneA & MaskW<W>::range(lo0 - 1, 64 * W - 1)).lowest() + 1;     // first model >= qLo[t-1] that is kept
         // lanes beyond Q are "kept" (valid==false -> bit clear in pruneA), which mirrors running past the chain
         if (s < 1 || s > hi1) { err = 1; break; }
         if (s < lo1) s = lo1;
         int e = (hi0 < Q) ? hi0 + 1 : hi0;
         // MaxModelProb(e, t-1, minq = s): tee predecessors above the start point add alpha_N + beta_N of the model
         // before them (HFB.c:667-672).  exitSum of model q-2 arrives by shuffle.
         // exitSumLast = alpha_N(t-1) + beta_N(t-1) of this model (LZERO outside the beta beam), kept from step t-1
         const double ex2 = g.upBy2(exitSumLast);
         double mB = mmpA;
         if (valid && q >= 3 && (q - 1) > s && a1N_p1 > (float)LSMALL && ex2 > mB) mB = ex2;
         const MaskW<W> pruneB = g.ballot(valid && (pr - mB > minF));
         e = (~pruneB & MaskW<W>::range(0, e - 1) & MaskW<W>::range(0, Q - 1)).highest() + 1;
         if (e < 1 || e < s) { err = 1; break; }
         while (e < Q && teeMask.bit(e - 1)) e++;
         if (e > hi1) e = hi1;
         sq = s; eq = e;
         // ---- alpha column t (HFB.c:729-771)
#pragma unroll
         for (int i = 0; i <= MAXN; i++) aP[i] = aC[i];
         double exP = LZERO;                             // alpha_N(t-1) of this model
#pragma unroll
         for (int i = 1; i <= MAXN; i++) if (i == N) exP = aP[i];
         double exP1, exP2;
         g.up2(exP, exP1, exP2);
         if (valid) {
            if (q < sq || q > eq) {
#pragma unroll
               for (int i = 1; i <= MAXN; i++) aC[i] = LZERO;
            } else {
               double a1;
               if (q == 1) a1 = LZERO;
               else {
                  a1 = exP1;
                  if (q > sq && a1N_p1 > (float)LSMALL) {     // previous model is a tee model
                     const double a1p = (q - 1 == 1) ? LZERO : exP2;
                     a1 = ladd(a1, a1p + (double)a1N_p1);
                  }
               }
               aC[1] = a1;
#pragma unroll
               for (int j = 2; j < MAXN; j++)
                  if (j < N) {
                     double aa = m.tp[0][j - 1];
                     double x = (aa > LSMALL) ? aa + a1 : LZERO;
#pragma unroll
                     for (int i = 2; i < MAXN; i++)
                        if (i < N) {
                           aa = m.tp[i - 1][j - 1];
                           const double y = aP[i];
                           if (aa > LSMALL && y > LSMALL) x = ladd(x, y + aa);
                        }
                     xpre[j] = x;
                     aC[j] = x + (double)oT[j];
                  }
               double x = LZERO;
#pragma unroll
               for (int i = 2; i < MAXN; i++)
                  if (i < N) {
                     const double aa = m.aN[i - 1], y = aC[i];
                     if (aa > LSMALL && y > LSMALL) x = ladd(x, y + aa);
                  }
#pragma unroll
               for (int i = 1; i <= MAXN; i++) if (i == N) aC[i] = x;
            }
         }
      }
      if (gl == 0) { gaLo[t] = (short)sq; gaHi[t] = (short)eq; }
      if (a.alphaDbg && valid) {
#pragma unroll
         for (int i = 1; i <= MAXN; i++) if (i <= N) a.alphaDbg[ud.beta0 + (size_t)(t - 1) * nC + m.mc0 + i - 1] = aC[i];
      }

      // ---- statistics for column t (HFB.c:1790-1806) and MaxModelProb of column t
      const double bEntryNext = g.down1(bT[1]);          // beta_1(q+1, t)
      double exitSumT = LZERO;                           // alpha_N(t) + beta_N(t) if this model is in the beta beam of t
      const bool inB = valid && q >= lo1 && q <= hi1;
      if (inB) {
#pragma unroll
         for (int i = 1; i <= MAXN; i++) if (i == N) exitSumT = aC[i] + bT[i];
      }
      const double exitSumPrev = g.up1(exitSumT);
      {
         double mm = (q > 1) ? exitSumPrev : LZERO;      // HFB.c:662-666
         if (inB) {
#pragma unroll
            for (int i = 1; i < MAXN; i++)
               if (i < N) { const double x = aC[i] + bT[i]; if (x > mm) mm = x; }
         }
         mmpA = mm;
         exitSumLast = exitSumT;
      }
      if (valid) {
         const bool inBeam = q >= sq && q <= eq;
         const bool bqt1ok = (t < T) && q >= lo2 && q <= hi2;
         const bool bq1tok = (q < Q) && (q + 1) >= lo1 && (q + 1) <= hi1;
         double bN = LZERO;
#pragma unroll
         for (int i = 1; i <= MAXN; i++) if (i == N) bN = bT[i];
#pragma unroll
         for (int i = 1; i < MAXN; i++) {
            if (i < N && inBeam) {
               const double ai = aC[i], bi = bT[i];
               // SetOcct (HFB.c:399-418)
               double x = ai + bi;
               if (i == 1 && bq1tok && a1N > (float)LSMALL) x = ladd(x, ai + bEntryNext + (double)a1N);
               x -= pr;
               const float occ = (x > EXPFLOOR) ? (float)EXPT(x) : 0.0f;
               occAcc[i] += (double)occ;
               if (wantTrans) {                          // UpTranParms (HFB.c:1390-1410), row i
                  if (i == 1) {
#pragma unroll
                     for (int j = 2; j < MAXN; j++)
                        if (j < N) {
                           x = ai + (double)m.tp[0][j - 1] + (double)oT[j] + bT[j] - pr;
                           if (x > EXPFLOOR) ta[1][j] += EXPT(x);
                        }
                     if (a1N > (float)LSMALL && bq1tok) {
                        x = ai + (double)a1N + bEntryNext - pr;
                        if (x > EXPFLOOR) {
                           const double e = EXPT(x);
#pragma unroll
                           for (int j = 2; j <= MAXN; j++) if (j == N) ta[1][j] += e;
                        }
                     }
                  } else {
                     if (bqt1ok) {
#pragma unroll
                        for (int j = 2; j < MAXN; j++)
                           if (j < N) {
                              x = ai + (double)m.tp[i - 1][j - 1] + (double)oT1[j] + bT1[j] - pr;
                              if (x > EXPFLOOR) ta[i][j] += EXPT(x);
                           }
                     }
                     x = ai + (double)m.aN[i - 1] + bN - pr;
                     if (x > EXPFLOOR) {
                        const double e = EXPT(x);
#pragma unroll
                        for (int j = 2; j <= MAXN; j++) if (j == N) ta[i][j] += e;
                     }
                  }
               }
            }
         }
         // UpMixParms seeds (HFB.c:1479-1489,1573-1606), one per emitting state
#pragma unroll
         for (int j = 2; j < MAXN; j++)
            if (j < N) {
               double seed = LZERO;
               if (inBeam && wantMix) {
                  if (cM[j] == 1 || a.maxM == 1) {
                     const double x = aC[j] + bT[j] - pr;
                     if (-x < minF) seed = x;
                  } else {
                     // initx: the alpha recursion's sum before b_j(t) was added (same operands, same order)
                     const double initx = xpre[j] + (bT[j] - pr);
                     const double ub = initx + (double)oT[j];
                     if (ub > -minF - 0.01) seed = initx;
                  }
               }
               gam[(size_t)(t - 1) * nSlots + j - 2] = seed;
            }
      }
      // rotate the rings: t -> t+1
#pragma unroll
      for (int i = 0; i <= MAXN; i++) { bT[i] = bT1[i]; bT1[i] = bT2[i]; }
#pragma unroll
      for (int j = 0; j < MAXN; j++) { oT[j] = oT1[j]; oT1[j] = oT2[j]; }
      lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2; lo2 = lo3; hi2 = hi3;
   }

   if (err) {
      if (gl == 0) { a.status[u] = HTKAMD_UTT_EALPHA; atomicAdd(a.acc + a.lay.nUttSkipped, 1.0); }
      return;
   }
   // ---- flush the per-model sums.  Models of an utterance usually share their transition matrix (always, in a
   // tied-transition system): then the counts are first summed over the lanes, so that an utterance issues ONE atomic
   // per matrix entry -- same-address f64 atomics serialise in L2, and 41 lanes x 1250 utterances on 15 addresses
   // used to cost more than the whole recursion.  (Each wavefront of a multi-wave utterance flushes its own 64 models.)
   if (wantTrans) {
      const int t0 = __shfl(cTrans, 0);
      const bool uniform = __all(!valid || cTrans == t0);
      if (uniform) {
         const int N0 = __shfl(N, 0);
         double *tr = a.acc + a.lay.tr + a.transOff[t0];
#pragma unroll
         for (int i = 1; i < MAXN; i++) {
#pragma unroll
            for (int j = 2; j <= MAXN; j++) {
               double v = (valid && i < N && j <= N) ? ta[i][j] : 0.0;
#pragma unroll
               for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
               if (lane == 0 && i < N0 && j <= N0 && v != 0.0) atomicAdd(tr + (size_t)(i - 1) * N0 + (j - 1), v);
            }
            double v = (valid && i < N) ? occAcc[i] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && i < N0 && v != 0.0) atomicAdd(a.acc + a.lay.trOcc + a.trOccOff[t0] + (i - 1), v);
         }
      } else if (valid) {
         double *tr = a.acc + a.lay.tr + a.transOff[cTrans];
#pragma unroll
         for (int i = 1; i < MAXN; i++)
            if (i < N) {
#pragma unroll
               for (int j = 2; j <= MAXN; j++)
                  if (j <= N && ta[i][j] != 0.0) atomicAdd(tr + (size_t)(i - 1) * N + (j - 1), ta[i][j]);
               if (occAcc[i] != 0.0) atomicAdd(a.acc + a.lay.trOcc + a.trOccOff[cTrans] + (i - 1), occAcc[i]);
            }
      }
   }
   if (valid) atomicAdd(a.acc + a.lay.nEgs + cHmm, 1.0);
   if (gl == 0) {
      atomicAdd(a.acc + a.lay.totalPr, pr);
      atomicAdd(a.acc + a.lay.totalT, (double)T);
      atomicAdd(a.acc + a.lay.nUttDone, 1.0);
      atomicAdd(a.acc + a.lay.nEval, (double)ud.nEval);
   }
}

// a.uttList / a.nList: the utterances of one class (W wavefronts each)
template <bool FAST> static void launch_beta_w(const FbArgs &a, int W, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_beta_w<5, 1, FAST>), dim3((a.nList + 3) / 4), dim3(256), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_beta_w<5, 2, FAST>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_beta_w<5, 4, FAST>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_beta_w<5, 8, FAST>), dim3(a.nList), dim3(512), 0, s, a);
}
template <bool FAST> static void launch_alpha_w(const FbArgs &a, int W, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_alpha_w<5, 1, FAST>), dim3((a.nList + 3) / 4), dim3(256), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_alpha_w<5, 2, FAST>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_alpha_w<5, 4, FAST>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_alpha_w<5, 8, FAST>), dim3(a.nList), dim3(512), 0, s, a);
}

// fast: tolerance-class LAdd / exp (ladd.h); the exact forms otherwise
int htkamd_launch_beta_w(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (fast) launch_beta_w<true>(a, W, s); else launch_beta_w<false>(a, W, s);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_launch_alpha_w(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (fast) launch_alpha_w<true>(a, W, s); else launch_alpha_w<false>(a, W, s);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}
