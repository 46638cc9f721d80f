// fb_lr.hip -- K2r / K3r / K3x: beta pass, alpha pass and occupation statistics for LEFT-TO-RIGHT chains, one lane per chain state.
//
// The everyday HMM set -- every model left-to-right without skips: the entry state reaches state 2 only, emitting state i reaches
// i and i+1 only, the last emitting state alone reaches the exit state, no tee model -- is a special case of what fb_state.hip
// handles, and special enough to be worth its own kernels (round 3):
//
//  * a state has ONE predecessor besides itself (alpha) and ONE successor besides itself (beta), and the neighbour is the lane next
//    door whether it belongs to the same model or to the model before / after: a step is one log-add per lane, one LDS exchange;
//  * the entry-state values need no storage: beta_1(q,t) = a_12 + b_2(t) + beta_2(t) is two additions from what the model's first
//    lane holds anyway (fb_state.hip writes and reads a second fp64 column for it: 1.95x the algorithmic beta traffic);
//  * NOTHING THAT DOES NOT FEED THE NEXT STEP IS IN THE CHAIN.  k_alpha_s (fb_state.hip) spends ~60 % of a step's instructions on
//    occupation counts, transition counts and mixture seeds inside its 500-step dependent chain; here the alpha kernel stores its
//    column (fp64, like beta) and a FRAME-PARALLEL kernel (k_stats_lr: workgroup = utterance x 64 frames, no dependence between
//    frames, ten thousand workgroups instead of 1 250) computes SetOcct / UpTranParms / the UpMixParms seeds from the stored alpha,
//    beta and scores.  Its per-workgroup transition counts go to a table of partial rows that k_trans_reduce sums (one atomic per
//    matrix entry and 256 rows: atomics of 10 000 workgroups on the 15 addresses of a tied matrix would serialise in L2).
//
// Reference semantics as fb_state.hip: SetBeta HFB.c:1149, StepAlpha :686, InitAlpha :616, MaxModelProb :655, SetOcct :399,
// UpTranParms :1371, UpMixParms seeds :1479 (S == 1).  Arithmetic operand for operand: with FAST = false every alpha, beta and the
// utterance probability equal fb_state.hip's and the oracle's bit for bit; FAST = true is the tolerance class of ladd.h.
//
// Layout per utterance (L = 64 W lanes, T frames, Q models; QP = Q rounded up to 8):
//   betaW [betaW0  + (t-1) L + lane]                     beta_j(t) of the lane's state (inside the beta beam of t)
//   alphaW[alphaW0 + (t-1) L + lane]                     alpha_j(t) - b_j(t)  ("xpre": what the mixture seeds need; alpha_j = xpre + b_j exactly)
//   alphaW[alphaW0 + T L + (t-1) QP + q-1]               alpha_1(q,t), the entry-state value of model q
//   qBeam / aBeam [frame0 + t-1]                         lo | hi << 16 of the beta and alpha beams (one scalar load per frame)
#include <hip/hip_runtime.h>
#include <type_traits>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "ladd.h"
#include "wavegrp.h"
#include "fb_state.h"

typedef const __attribute__((address_space(4))) int cint_lr;

#define ALPHA_S(t) (a.alphaW[ud.alphaW0 + (size_t)((t) - 1) * L + gl])
#define ALPHA_E(t, q_) (a.alphaW[ud.alphaW0 + (size_t)T * L + (size_t)((t) - 1) * ud.QP + ((q_) - 1)])

struct LrRegs {
   int q, j, N, mi;
   bool first, last;
   float aSelf, aNext, aPrev;          // a_jj, a_j,j+1 (inner lanes), a_j-1,j (all but a model's first lane)
   float aEntry, aExit;                // a_12 at the first lane, a_N-1,N at the last
   float aExitPrev, aEntryNext;        // first lane: a_N'-1,N' of the model before; last lane: a_12 of the model after
};

__device__ __forceinline__ void load_lr(LrRegs &s, const FbArgs &a, const UttDesc &ud, int gl, bool valid)
{
   s.q = 0; s.j = 2; s.N = 3; s.mi = 0; s.first = false; s.last = false;
   s.aSelf = s.aNext = s.aPrev = s.aEntry = s.aExit = s.aExitPrev = s.aEntryNext = (float)LZERO;
   if (!valid) return;
   s.q = a.sQ[ud.slot0 + gl];
   s.mi = ud.q0 + s.q - 1;
   s.N = a.mN[s.mi];
   s.j = gl - a.mSlot0[s.mi] + 2;
   const float *tp = a.transP + a.mTp[s.mi];
   const int N = s.N, j = s.j;
   s.first = j == 2; s.last = j == N - 1;
   s.aSelf = tp[(j - 1) * N + (j - 1)];
   if (!s.last) s.aNext = tp[(j - 1) * N + j];
   if (!s.first) s.aPrev = tp[(j - 2) * N + (j - 1)];
   if (s.first) s.aEntry = tp[1];
   if (s.last) s.aExit = tp[(j - 1) * N + (N - 1)];
   if (s.first && s.q > 1) { const int mp = s.mi - 1, Np = a.mN[mp]; s.aExitPrev = a.transP[a.mTp[mp] + (Np - 2) * Np + (Np - 1)]; }
   if (s.last && s.q < ud.Q) s.aEntryNext = a.transP[a.mTp[s.mi + 1] + 1];
}

// what the sparse statistics kernel (k_stats_sp) needs of a chain state, left by the beta kernels in the batch's record table
__device__ __forceinline__ void store_lane_rec(const FbArgs &a, const UttDesc &ud, int gl, bool valid, const LrRegs &s)
{
   if (!valid || !a.laneRec) return;
   LaneRec r;
   r.aSelf = s.aSelf; r.aOut = s.last ? s.aExit : s.aNext; r.aEntry = s.aEntry; r.aEntryNext = s.aEntryNext;
   r.q = (short)s.q; r.j = (short)s.j; r.N = (short)s.N; r.pad = 0;
   r.sidx = a.slotState[ud.slot0 + gl];
   r.cM = a.stateCompOff[r.sidx + 1] - a.stateCompOff[r.sidx];
   a.laneRec[ud.slot0 + gl] = r;
}

// LAdd(LZERO, v) of the reference (HMath.c:1576): v itself unless it is below LSMALL
__device__ __forceinline__ double from_zero(double v) { return (v < LSMALL) ? LZERO : v; }

// entry-state beta of a model from its first state's values (SetBeta HFB.c:1232-1243 with one entry transition): a_12 + b_2 + beta_2
template <bool FAST> __device__ __forceinline__ double entry_beta(float aEntry, double o, double b)
{
   const double aa = aEntry;
   if constexpr (FAST) return aa + o + b;
   else return (aa > LSMALL && b > LSMALL) ? from_zero(aa + o + b) : LZERO;
}

__device__ __forceinline__ int beam_word(const int *p, int i) { return __builtin_amdgcn_readfirstlane(p[i]); }

// ------------------------------------------------------------------------------------ K2r: beta
template <int W, bool FAST>
__global__ __launch_bounds__(64 * W) void k_beta_lr(FbArgs a)
{
   constexpr int L = 64 * W, LP = L + 2 * SPAD;
   __shared__ double ltab[FAST ? 1 : LADD_TAB_DOUBLES];
   __shared__ double xbeta[2][LP];                     // beta_j of the column just computed, by step parity
   __shared__ double xobs[2][LP];                      // b_j of that column
   __shared__ float stage[W][2 * 64 * 4];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   __shared__ short sqOf[L];
   __shared__ short flOf[L + 2];
   if constexpr (!FAST) ladd_table_to_lds(ltab, a.laddTab);
   const int gl = threadIdx.x, lane = gl & 63, wv = gl >> 6;
   const int li = blockIdx.x;
   if (li >= a.nList) return;
   const int u = a.uttList[li];
   const UttDesc ud = a.utt[u];
   if (ud.status != HTKAMD_UTT_OK) {
      if (gl == 0) { a.status[u] = ud.status; a.pr[u] = LZERO; }
      return;
   }
   Grp<W> g; g.x = gx; g.a1 = ga1; g.wave = wv; g.lane = lane; g.ph = 0;
   const int T = ud.T, Q = ud.Q, nS = ud.nSlots;
   const bool valid = gl < nS;
   LrRegs s;
   load_lr(s, a, ud, gl, valid);
   store_lane_rec(a, ud, gl, valid, s);
   for (int i = gl; i < LP; i += L) { xbeta[0][i] = LZERO; xbeta[1][i] = LZERO; xobs[0][i] = 0.0; xobs[1][i] = 0.0; }
   sqOf[gl] = (short)(valid ? s.q : Q + 1);
   if (valid && s.first) flOf[s.q] = (short)gl;
   if (gl == 0) { flOf[Q + 1] = (short)nS; flOf[0] = 0; }
   __syncthreads();
   const int q = s.q, N = s.N;
   const short *tLo = a.taperLo + ud.frame0 - 1, *tHi = a.taperHi + ud.frame0 - 1;   // 1-based t
   short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
   int *gBeam = a.qBeam + ud.frame0 - 1;
   ObsRow st;
   st.lds = stage[wv]; st.lane = lane; st.R = (f4s)(0.f);
   st.row = a.outp + ud.outp0 + (size_t)(valid ? gl : 0) * T;      // (lanes past the chain read the first state's row: no lane-divergent loads)
   const double mle = a.minLogExp;
   const bool pruning = a.pruneInit < HTKAMD_NOPRUNE;
#define ladd(x, y) ladd_sel<FAST>((x), (y), mle, ltab)

   double thresh = a.pruneInit, pr = LZERO;
   int ok = 0;
   for (;;) {                                            // StepBack retry loop (HFB.c:1332-1361)
      int fail = 0;
      double bJ = LZERO;                                 // beta_j of the column last computed
      double yN = LZERO, oN = 0.0;                       // ... and beta, score of the lane next door in that column
      double lMax = LZERO;
      float obT = 0.f, obP = 0.f;
      double obL = 0.0;                                  // own score of the column last computed
      // ---- t = T (HFB.c:1175-1198): only the last model can end the utterance
      const int endT = tLo[T];
      {
         const int bl = (T - 1) >> 2;
         st.load_all(bl); st.park(bl);
         if (bl >= 1) st.load_all(bl - 1);
         obT = st.get_all(T - 1);
         if (T >= 2) {
            if (((T - 2) & 3) == 3) { st.park((T - 2) >> 2); if (((T - 2) >> 2) >= 1) st.load_all(((T - 2) >> 2) - 1); }
            obP = st.get_all(T - 2);
         }
      }
      const bool inT = valid && q >= endT;
      {
         double mine = 0.0;
         for (int k = Q; k > q && k > endT; k--) mine += (double)(float)LZERO;
         if (inT) bJ = (double)s.aExit + mine;
         obL = (double)obT;
         xbeta[T & 1][SPAD + gl] = inT ? bJ : LZERO; xobs[T & 1][SPAD + gl] = obL;
         xsync<W>();
         yN = xbeta[T & 1][SPAD + gl + 1]; oN = xobs[T & 1][SPAD + gl + 1];
         if (inT) BETA_S(T) = bJ;
      }
      if (gl == 0) { gLo[T] = (short)endT; gHi[T] = (short)Q; gBeam[T] = endT | (Q << 16); }
      int qHiN = Q, qLoN = endT, lastEnd = endT;
      int nxtLo = (T >= 2) ? tLo[T - 1] : 1, nxtHi = (T >= 2) ? tHi[T - 1] : 1;
      bool stPrev = false, stIn = false; int tPrev = 0, loPrev = 1, hiPrev = 1;

      // ---- t = T-1 .. 1 (HFB.c:1205-1277)
      for (int t = T - 1; t >= 1; t--) {
         const int taperLoT = nxtLo, taperHiT = nxtHi;
         if (t >= 2) { nxtLo = tLo[t - 1]; nxtHi = tHi[t - 1]; }
         if (t >= 2 && ((t - 2) & 3) == 3) { st.park((t - 2) >> 2); if (((t - 2) >> 2) >= 1) st.load_all(((t - 2) >> 2) - 1); }
         if (stPrev) {                                   // the column finished in the previous iteration goes out now
            if (stIn) BETA_S(tPrev) = bJ;
            if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; gBeam[tPrev] = loPrev | (hiPrev << 16); }
            stPrev = false;
         }
         obT = obP;
         if (t >= 2) obP = st.get_all(t - 2);
         const int startq = qHiN;
         const int endq = (qLoN == 1) ? 1 : ((taperLoT >= qLoN) ? taperLoT : qLoN - 1);
         const bool inRange = valid && q >= endq && q <= startq;
         const bool wasIn = q >= qLoN && q <= qHiN;
         if constexpr (FAST) {                          // (every lane, selects instead of a lane-divergent branch)
            const bool p1 = (q < Q) && (q + 1 >= qLoN) && (q + 1 <= qHiN);
            // the lane next door: the next state of the model, or (last lane) the exit state = the entry state of the next model
            const double nb = s.last ? ((double)s.aExit + (p1 ? (double)s.aEntryNext + oN + yN : LZERO))
                                     : (wasIn ? (double)s.aNext + oN + yN : LZERO);
            const double self = wasIn ? (double)s.aSelf + obL + bJ : LZERO;
            const double bn = ladd_fast(nb, self);
            bJ = inRange ? bn : bJ;
         } else if (inRange) {
            const bool p1 = (q < Q) && (q + 1 >= qLoN) && (q + 1 <= qHiN);
            {
               double ex = LZERO;                                   // beta_N(q,t) = beta_1(q+1,t+1)
               if (s.last && p1) ex = entry_beta<false>(s.aEntryNext, oN, yN);
               double x = (double)s.aExit + ex;
               if (wasIn) {
                  double aa = s.aSelf;
                  if (aa > LSMALL && bJ > LSMALL) x = ladd(x, aa + obL + bJ);
                  aa = s.aNext;
                  if (aa > LSMALL && yN > LSMALL) x = ladd(x, aa + oN + yN);
               }
               bJ = x;
            }
         }
         // the one exchange of the step
         obL = (double)obT;
         xbeta[t & 1][SPAD + gl] = inRange ? bJ : LZERO; xobs[t & 1][SPAD + gl] = obL;
         xsync<W>();
         yN = xbeta[t & 1][SPAD + gl + 1]; oN = xobs[t & 1][SPAD + gl + 1];
         int newHi, newLo;
         if (!pruning) {                                 // only the taper acts (HFB.c:1259-1264)
            newHi = (taperHiT < startq) ? taperHiT : startq;
            newLo = endq;
         } else {                                        // beam pruning (HFB.c:1254-1272): one bit per model, at its first lane
            lMax = LZERO;
            if (inRange && s.first) {
               const double *xb_ = xbeta[t & 1] + SPAD + gl;
#pragma unroll
               for (int k = 0; k < 3; k++) if (2 + k <= N - 1) { const double y = xb_[k]; if (y > lMax) lMax = y; }
            }
            const bool rep = inRange && s.first;
            const double gmax = g.maxall(rep ? lMax : LZERO);
            const MaskW<W> keep = g.ballot(rep && !(gmax - lMax > thresh));
            const int sl = (keep & MaskW<W>::range(0, flOf[startq + 1] - 1)).highest();
            int sN = (sl >= 0) ? sqOf[sl] : 0;
            if (sN >= 1 && taperHiT < sN) sN = taperHiT;
            if (sN < 1) { fail = 1; newHi = newLo = 1; }
            else if (keep.bit(flOf[endq])) { newHi = sN; newLo = endq; }
            else {
               const int el = (keep & MaskW<W>::range(flOf[endq + 1], flOf[sN + 1] - 1)).lowest();
               if (el < 0) { fail = 1; newHi = newLo = 1; }
               else { newHi = sN; newLo = sqOf[el]; }
            }
         }
         if (fail) break;
         stPrev = true; stIn = inRange; tPrev = t; loPrev = newLo; hiPrev = newHi;
         qHiN = newHi; qLoN = newLo; lastEnd = endq;
      }
      if (!fail && stPrev) {                             // the last column (t = 1)
         if (stIn) BETA_S(tPrev) = bJ;
         if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; gBeam[tPrev] = loPrev | (hiPrev << 16); }
      }
      if (!fail) {
         // utt->pr = beta_1 of the last model processed, in the last column (own values of its first lane)
         const double bE = (valid && s.first) ? entry_beta<FAST>(s.aEntry, obL, bJ) : LZERO;
         pr = g.bcast(bE, flOf[lastEnd]);
         if (pr > LSMALL) { ok = 1; break; }
      }
      thresh += a.pruneInc;
      if (thresh > a.pruneLim || a.pruneInc == 0.0) break;
      __syncthreads();
   }
   if (gl == 0) {
      a.pr[u] = ok ? pr : LZERO;
      a.status[u] = ok ? HTKAMD_UTT_OK : HTKAMD_UTT_SKIPPED;
   }
#undef ladd
}

// ------------------------------------------------------------------------------------ K3r: alpha (the chain only)
// ONE workgroup barrier per step.  The alpha beam of column t (HFB.c:699-722) is decided by MaxModelProb of column t-1, one bit per
// model -- a ballot over the workgroup's wavefronts, i.e. an exchange of its own.  Here the ballot word of a wavefront travels with the
// column exchange of the SAME step: every lane computes its alpha_j(t) before the beam is known, publishes it unmasked together with
// the ballot, and after the barrier everybody derives the beam, masks its own value and masks what it reads of its neighbour (a lane
// knows the first lane of its neighbour's model, which is what the beam is expressed in).
template <int W, bool FAST>
__global__ __launch_bounds__(64 * W) void k_alpha_lr(FbArgs a)
{
   constexpr int L = 64 * W, LP = L + 2 * SPAD;
   __shared__ double ltab[FAST ? 1 : LADD_TAB_DOUBLES];
   __shared__ double xalpha[2][LP];                    // alpha_j(t) by step parity (unmasked for t >= 2)
   __shared__ double xsum[2][LP];                      // alpha_j(t) + beta_j(t) inside the beta beam (MaxModelProb)
   __shared__ unsigned long long bslot[2][8];          // the wavefronts' ballot words, by step parity
   __shared__ float stage[W][2 * 64 * 4];
   __shared__ short flOf[L + 2];
   if constexpr (!FAST) ladd_table_to_lds(ltab, a.laddTab);
   const int gl = threadIdx.x, lane = gl & 63, wv = gl >> 6;
   const int li = blockIdx.x;
   if (li >= a.nList) return;
   const int u = a.uttList[li];
   const UttDesc ud = a.utt[u];
   if (a.status[u] != HTKAMD_UTT_OK) {                   // skipped in the beta pass (or pre-check)
      if (gl == 0) atomicAdd(a.acc + a.lay.nUttSkipped, 1.0);
      return;
   }
   const int T = ud.T, Q = ud.Q, nS = ud.nSlots, nC = ud.nCells;
   const bool valid = gl < nS;
   LrRegs s;
   load_lr(s, a, ud, gl, valid);
   for (int i = gl; i < LP; i += L) {
#pragma unroll
      for (int k = 0; k < 2; k++) { xalpha[k][i] = LZERO; xsum[k][i] = LZERO; }
   }
   if (valid && s.first) flOf[s.q] = (short)gl;
   if (gl == 0) { flOf[Q + 1] = (short)nS; flOf[0] = 0; }
   {  // one bit per model, at its first lane (loop-invariant)
      const unsigned long long b = __ballot(valid && s.first);
      if (lane == 0) bslot[0][wv] = b;
   }
   __syncthreads();
   MaskW<W> firsts;
#pragma unroll
   for (int k = 0; k < W; k++) firsts.w[k] = bslot[0][k];
   __syncthreads();
   const int q = s.q, N = s.N, j = s.j;
   const int cHmm = valid ? a.mHmm[s.mi] : 0, mc0 = valid ? a.mCell0[s.mi] : 0;
   cint_lr *qBeam = (cint_lr *)(a.qBeam + ud.frame0 - 1);     // final beta beams, 1-based t, written by the beta launch before this one
   int *gaBeam = a.aBeam + ud.frame0 - 1;                      // alpha beams as FIRST LANES: first lane of model sq | first lane of model eq << 16
   ObsRow st;
   st.lds = stage[wv]; st.lane = lane; st.R = (f4s)(0.f);
   st.row = a.outp + ud.outp0 + (size_t)(valid ? gl : 0) * T;      // (lanes past the chain read the first state's row: no lane-divergent loads)
   const double mle = a.minLogExp, pr = a.pr[u];
   const double minF = (double)a.minFrwdP;
   const float aA = s.first ? s.aEntry : s.aPrev;        // the transition into the state from the lane before it / from the entry state
   const int myFirst = gl - (j - 2);                     // first lane of the own model
   // first lane of the model of the lane before this one (what its membership of the beam is decided by)
   int prevFirst = myFirst;
   if (valid && s.first && q > 1) prevFirst = flOf[q - 1];
#define ladd(x, y) ladd_sel<FAST>((x), (y), mle, ltab)

   double aJ = LZERO, aE = LZERO, aEnext = LZERO;        // alpha_j(t); alpha_1(q,t); alpha_1(q,t+1) = exit value of the model before, column t
   double yPrev = LZERO;                                 // alpha of the lane before this one in the column before (log-zero outside that column's beam)
   double xpre = LZERO;
   double bT = LZERO, bT1 = LZERO, bT2 = LZERO;          // beta of frames t, t+1 and (in flight) t+2
   float oT = 0.f, oT1 = 0.f;
   int w1 = qBeam[1], w2 = (T >= 2) ? qBeam[2] : 1, w3 = 1;
   int wq1 = w1, wq2 = w2, wSame = -1;                  // the words of columns t, t+1; their common value while the window t .. t+2 is uniform
   int hi0 = 0, lo1 = w1 & 0xffff, hi1 = w1 >> 16, lo2 = w2 & 0xffff, hi2 = (T >= 2) ? (w2 >> 16) : 0;
   double *pS = &ALPHA_S(1);                             // this lane's place in the stored columns, advanced by L per step (every lane of the workgroup has one)
   double *pE = (valid && s.first) ? &ALPHA_E(1, q) : nullptr;
   {
      bT = BETA_S(1);
      if (T >= 2) bT1 = BETA_S(2);
   }
   st.load_all(0); st.park(0);
   if (T > 4) st.load_all(1);
   oT = st.get_all(0); if (T >= 2) oT1 = st.get_all(1);
   int err = 0;
   double mmpA = LZERO;                                  // MaxModelProb of this model in the column just finished (first lane)
   int fLo0 = 0, fLo1 = flOf[lo1], fHi1 = flOf[hi1 > 0 ? hi1 : 1], fE0 = 0;
   // the two range masks of the beam decision, kept while their bounds stay (the beta beam's bounds move every few frames only)
   MaskW<W> mLo = MaskW<W>::range(0, L - 1), mE = MaskW<W>::range(0, -1);
   int mLoOf = 0, mEOf = 0, lo1Of = lo1, hi1Of = hi1, e0Of = -1;
   double eT = LZERO, eT1 = LZERO;                       // entry-state beta of the own model in columns t, t+1 (first lane)
   if (valid && s.first) { eT = entry_beta<FAST>(s.aEntry, (double)oT, bT); if (T >= 2) eT1 = entry_beta<FAST>(s.aEntry, (double)oT1, bT1); }

   // (column 1 -- InitAlpha -- is a call of its own: the loop body carries no branch on t and no copies of the state it does not touch)
   // (... and so are the last two columns: in the loop between, column t + 2 exists without asking)
   auto step = [&](const int t, auto first_, auto inner_) -> bool {
      constexpr bool FIRST = decltype(first_)::value, INNER = decltype(inner_)::value;
      const bool has2 = INNER || t + 2 <= T, isLast = !INNER && t == T;
      if (has2) {                                        // request column t+2 (lanes past the chain read their own unused places: no divergence)
         w3 = qBeam[t + 2];
         bT2 = BETA_S(t + 2);
      }
      const int par = t & 1;
      const bool inB = valid && q >= lo1 && q <= hi1;    // in the beta beam of t
      bool in;
      int sl = 0, el = 0;
      if constexpr (FIRST) {
         // ---- InitAlpha (HFB.c:616-651): without tee models only the first model starts
         const int eq = hi1;
         double a1 = 0.0;
         for (int k = 2; k <= q && k <= eq; k++) a1 += (double)(float)LZERO;
         in = valid && q <= eq;
         if (in) {
            aE = a1;
            const double aa = s.aEntry;
            xpre = aE + aa;
            aJ = (aa > LSMALL) ? xpre + (double)oT : LZERO;
         }
         sl = 0; el = fHi1;
         xalpha[par][SPAD + gl] = in ? aJ : LZERO;
         xsum[par][SPAD + gl] = inB ? aJ + bT : LZERO;
         xsync<W>();
         yPrev = xalpha[par][SPAD + gl - 1];
      } else {
         // ---- this wavefront's word of the ballot on MaxModelProb of column t-1
         const unsigned long long bal = __ballot(valid && s.first && (pr - mmpA > minF));
         if constexpr (W > 1) { if (lane == 0) bslot[par][wv] = bal; }
         // ---- alpha column t (HFB.c:729-771) as if the state were in the beam: entry term (first lane) or the state before (other
         // lanes), then the state itself
         const double a1 = (q == 1) ? LZERO : aEnext;                // alpha_1(q,t) = alpha_N(q-1,t-1)
         double x;
         if constexpr (FAST) {
            const double tA = (s.first ? a1 : yPrev) + (double)aA;
            x = ladd_fast(tA, aJ + (double)s.aSelf);
         } else {
            x = s.first ? (((double)s.aEntry > LSMALL) ? (double)s.aEntry + a1 : LZERO) : LZERO;
            if (!s.first && (double)s.aPrev > LSMALL && yPrev > LSMALL) x = ladd(x, yPrev + (double)s.aPrev);
            if ((double)s.aSelf > LSMALL && aJ > LSMALL) x = ladd(x, aJ + (double)s.aSelf);
         }
         const double aJn = x + (double)oT;
         xalpha[par][SPAD + gl] = aJn;
         xsum[par][SPAD + gl] = inB ? aJn + bT : LZERO;
         xsync<W>();
         // ---- alpha beam (HFB.c:699-722): the reference's comparisons of model numbers are made on the models' first lanes (F(x) =
         // flOf[x] is increasing in x); the bounds F(lo), F(hi) of the beta beams involved were looked up a step ahead
         MaskW<W> kept;
         if constexpr (W == 1) kept.w[0] = firsts.w[0] & ~bal;
         else {
#pragma unroll
            for (int k = 0; k < W; k++) kept.w[k] = firsts.w[k] & ~bslot[par][k];
         }
         const int slane = (kept & mLo).lowest();                                      // first model >= qLo[t-1] that is kept
         if (slane < 0 || slane > fHi1) { err = 1; return false; }                            // sq > qHi[t]
         sl = (slane < fLo1) ? fLo1 : slane;                                           // start-point below the beta beam: pulled back
         const int elane = (kept & mE).highest();                                      // last kept model <= min(qHi[t-1] + 1, Q)
         if (elane < 0 || elane < sl) { err = 1; return false; }
         el = (elane > fHi1) ? fHi1 : elane;
         in = valid && myFirst >= sl && myFirst <= el;
         aJ = in ? aJn : LZERO; aE = in ? a1 : LZERO; xpre = x;
         const bool inPrev = prevFirst >= sl && prevFirst <= el;
         const double yp = xalpha[par][SPAD + gl - 1];
         yPrev = inPrev ? yp : LZERO;
      }
      if (gl == 0) gaBeam[t] = sl | (el << 16);
      *pS = xpre; pS += L;
      if (valid && s.first) { *pE = aE; pE += ud.QP; }
      // exit value of the model BEFORE this one in column t (HFB.c:762-769 there): alpha_1 of this model in column t+1
      double aXp;
      if constexpr (FAST) aXp = yPrev + (double)s.aExitPrev;
      else aXp = ((double)s.aExitPrev > LSMALL && yPrev > LSMALL) ? from_zero(yPrev + (double)s.aExitPrev) : LZERO;
      if (a.alphaDbg && valid) {
         double *ad = a.alphaDbg + ud.beta0 + (size_t)(t - 1) * nC + mc0;
         ad[j - 1] = aJ;
         if (s.first) ad[0] = aE;
         if (s.last) {                                   // debugging aid only: this model's own exit value
            const double aa = s.aExit;
            ad[N - 1] = (aa > LSMALL && (t == 1 || aJ > LSMALL)) ? ladd(LZERO, aJ + aa) : LZERO;
         }
      }
      // ---- MaxModelProb of column t (HFB.c:655-683), at the model's first lane
      {  // (every lane computes it, selects instead of branches; the ballot of the next step asks the models' first lanes only)
         // (outside the alpha beam every term is a log-zero plus something: the model is not kept, whatever the sum)
         double mm = aE + eT;                            // i = 1
         const double *xs = xsum[par] + SPAD + gl;
#pragma unroll
         for (int k = 0; k < 3; k++) { const double v = xs[k]; mm = (2 + k <= N - 1 && v > mm) ? v : mm; }
         mm = (inB && in) ? mm : LZERO;                  // the published sums were not masked by the alpha beam
         // alpha_N + beta_N of the model before; its beta_N(t) is this model's beta_1(t+1) inside the beam of t+1
         const double bNp = (!isLast && q >= lo2 && q <= hi2) ? eT1 : LZERO;
         const double prevExit = (q > 1 && q - 1 >= lo1 && q - 1 <= hi1) ? aXp + bNp : LZERO;
         mmpA = (prevExit > mm) ? prevExit : mm;
      }
      aEnext = aXp;
      // rotate: t -> t+1
      bT = bT1; bT1 = bT2;
      oT = oT1;
      if (has2) {
         const int f = t + 1;                            // frame index (0-based) of t+2
         if ((f & 3) == 0) { st.park(f >> 2); if (4 * ((f >> 2) + 1) < T) st.load_all((f >> 2) + 1); }
         oT1 = st.get_all(f);
      }
      eT = eT1;
      if (has2) eT1 = entry_beta<FAST>(s.aEntry, (double)oT1, bT1);      // (meaningful at the models' first lanes)
      // The beta beam's words of columns t, t+1, t+2 decide everything below, and without pruning they stay the same for hundreds of
      // columns: while the word coming in equals the one all three had (wSame), nothing moves and the step skips the lot.
      if (!INNER || w3 != wSame) {
         hi0 = hi1; lo1 = lo2; hi1 = hi2; lo2 = w3 & 0xffff; hi2 = has2 ? (w3 >> 16) : 0;
         // first lanes of the models that bound the next step's beam decisions (hi1 may be 0 past the last frame: the step is not taken)
         fLo0 = fLo1;
         if (lo1 != lo1Of) { fLo1 = flOf[lo1]; lo1Of = lo1; }
         if (hi1 != hi1Of) { fHi1 = flOf[hi1 > 0 ? hi1 : 1]; hi1Of = hi1; }
         { const int e0 = ((hi0 < Q) ? hi0 + 1 : hi0) + 1; if (e0 != e0Of) { fE0 = flOf[e0]; e0Of = e0; } }
         if (fLo0 != mLoOf) { mLo = MaskW<W>::range(fLo0, L - 1); mLoOf = fLo0; }
         if (fE0 != mEOf) { mE = MaskW<W>::range(0, fE0 - 1); mEOf = fE0; }
         wSame = (has2 && wq1 == wq2 && wq2 == w3) ? w3 : -1;
         wq1 = wq2; wq2 = w3;
      }
      return true;
   };
   if (T >= 1 && step(1, std::true_type{}, std::false_type{})) {
      int t = 2;
      for (; t + 2 <= T; t++) if (!step(t, std::false_type{}, std::true_type{})) { t = T + 1; break; }
      for (; t <= T; t++) if (!step(t, std::false_type{}, std::false_type{})) break;
   }

   if (err) {
      if (gl == 0) { a.status[u] = HTKAMD_UTT_EALPHA; atomicAdd(a.acc + a.lay.nUttSkipped, 1.0); }
      return;
   }
   if (valid && s.first) atomicAdd(a.acc + a.lay.nEgs + cHmm, 1.0);
   if (gl == 0) {
      atomicAdd(a.acc + a.lay.totalPr, pr);
      atomicAdd(a.acc + a.lay.totalT, (double)T);
      atomicAdd(a.acc + a.lay.nUttDone, 1.0);
      atomicAdd(a.acc + a.lay.nEval, (double)ud.nEval);
   }
#undef ladd
}

// ------------------------------------------------------------------------------------ K3x: occupation / transition counts, mixture seeds
// Workgroup = (utterance, chunk of STATS_FC frames), lane = chain state as above.  Per frame and lane: SetOcct (HFB.c:399-418),
// UpTranParms (HFB.c:1390-1410) and the UpMixParms seed (HFB.c:1479-1489,1573-1606) from the stored alpha column, beta column and
// scores; frames are independent, the loads of the frames of a chunk are all in flight together.
#ifndef STATS_FC
#define STATS_FC 32
#endif
#define TR_ROW 16
#define EXP_TERM(acc, x) do { if constexpr (FAST) acc += exp_fast(x); else if ((x) > EXPFLOOR) acc += exp_tab((x), etab); } while (0)

template <int W, bool FAST>
__global__ __launch_bounds__(64 * W) void k_stats_lr(FbArgs a)
{
   constexpr int L = 64 * W, FC = STATS_FC, OS = FC + 3;
   __shared__ double etab[FAST ? 1 : EXP_TAB_N];
   __shared__ float otile[(L + 1) * OS];                 // scores of frames t0 .. t1+1, one row per lane (+ one row of zeros after the last)
   // the surviving pairs of a wavefront wait here and go out in whole lines at the end (a global store per frame made the next frame wait
   // for it: the compiler drains vmcnt before it reuses the store's registers -- 0.33 ms of the kernel's 0.59)
   constexpr int HB = 128;
   __shared__ int hbSt[W][HB], hbFr[W][HB];
   __shared__ double hbSeed[W][HB];
   if constexpr (!FAST) exp_table_to_lds(etab);
   const int gl = threadIdx.x, lane = gl & 63, wv = gl >> 6;
   const int li = blockIdx.x, ch = blockIdx.y;
   const size_t region = (size_t)((size_t)li * gridDim.y + ch) * W + wv;      // this wavefront's row of partial counts and region of the hit list
   double *part = a.trPart + region * TR_ROW;
   const int u = a.uttList[li];
   const UttDesc ud = a.utt[u];
   const int T = ud.T, Q = ud.Q, nS = ud.nSlots;
   const int t0 = ch * FC + 1;
   if (a.status[u] != HTKAMD_UTT_OK || t0 > T) {         // skipped, failed in the alpha pass, or shorter than this chunk's first frame
      if (lane == 0) { part[TR_ROW - 1] = -1.0; a.hitCtl[region] = 0; }
      return;
   }
   const int t1 = (t0 + FC - 1 < T) ? t0 + FC - 1 : T;
   const bool valid = gl < nS;
   LrRegs s;
   load_lr(s, a, ud, gl, valid);
   const int q = s.q;
   const int myFirst = gl - (s.j - 2);
   // scores: the lane's own row, frames t0 .. min(T, t1+1)
   {
      const int nf = ((t1 + 1 < T) ? t1 + 1 : T) - t0 + 1;
      float *dst = otile + gl * OS;
      if (valid) {
         const float *row = a.outp + ud.outp0 + (size_t)gl * T + (t0 - 1);
         int k = 0;
         for (; k + 4 <= nf; k += 4) { const f4s v = *(const f4s *)(row + k); dst[k] = v[0]; dst[k + 1] = v[1]; dst[k + 2] = v[2]; dst[k + 3] = v[3]; }
         for (; k < nf; k++) dst[k] = row[k];
      } else for (int k = 0; k < nf; k++) dst[k] = 0.0f;
      if (gl == L - 1) for (int k = 0; k < OS; k++) otile[L * OS + k] = 0.0f;
   }
   __syncthreads();
   cint_lr *qBeam = (cint_lr *)(a.qBeam + ud.frame0 - 1);
   cint_lr *aBeam = (cint_lr *)(a.aBeam + ud.frame0 - 1);
   const double pr = a.pr[u], minF = (double)a.minFrwdP;
   const bool wantMix = (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES)) != 0;
   const bool wantTrans = (a.uFlags & HTKAMD_UPTRANS) != 0;
   int cM = 0, sidx = 0;
   if (valid) { sidx = a.slotState[ud.slot0 + gl]; cM = a.stateCompOff[sidx + 1] - a.stateCompOff[sidx]; }
   MixHit *hreg = a.hits + region * (size_t)(STATS_FC * 64);
   int hc = 0, hb = 0;                                   // records in this wavefront's region of the hit list / waiting in its LDS buffer
   const bool single = (a.maxM == 1);
   const bool hasNext = valid && q < Q;
   const bool nbValid = gl + 1 < nS;                     // the lane next door holds a state
   double taSelf = 0.0, taOut = 0.0, taEntry = 0.0, occJ = 0.0, occE = 0.0;       // taOut: to the next state (inner lanes) / to the exit state (last lane)
   const float *orow = otile + gl * OS;

   // four frames at a time: every load of the four is issued before the first of them is used (the stores of the seeds would otherwise
   // keep the next frame's loads behind them)
   for (int tb = t0; tb <= t1; tb += 4) {
      double xpv[4], aEv[4], bv[5], bnv[4];
      int wav[4], wbv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const int t = tb + k;
         xpv[k] = LZERO; aEv[k] = LZERO; bnv[k] = LZERO; wav[k] = 1; wbv[k] = 1;
         if (t <= t1) {
            wav[k] = aBeam[t];
            if (t < T) wbv[k] = qBeam[t + 1];
            if (valid) {
               xpv[k] = ALPHA_S(t);
               if (s.first) aEv[k] = ALPHA_E(t, q);
               // beta_{j+1}(t+1): the lane next door's own bv[k + 1] -- taken from it below (a lane shift); only a wavefront's last lane,
               // whose neighbour sits in the next wavefront, reads it from memory (round 3: every lane did -- the column was read twice)
               if (lane == 63 && t < T && nbValid) bnv[k] = a.betaW[ud.betaW0 + (size_t)t * L + gl + 1];
            }
         }
      }
#pragma unroll
      for (int k = 0; k < 5; k++) { const int t = tb + k; bv[k] = (valid && t <= T && t <= t1 + 1) ? BETA_S(t) : LZERO; }
#pragma unroll
      for (int k = 0; k < 4; k++) {
         const double nb = __shfl_down(bv[k + 1], 1);      // (an invalid or out-of-range neighbour holds LZERO there, as the load gave)
         if (lane != 63) bnv[k] = (tb + k <= t1) ? nb : LZERO;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
      const int t = tb + k;
      if (t > t1) break;
      const int sl = wav[k] & 0xffff, el = wav[k] >> 16, lo2 = wbv[k] & 0xffff, hi2 = (t < T) ? (wbv[k] >> 16) : 0;
      const bool inBeam = valid && myFirst >= sl && myFirst <= el;          // the alpha beam comes as the first lanes of its first and last model
      const bool bqt1ok = (t < T) && q >= lo2 && q <= hi2;
      const double xp = xpv[k], aE = aEv[k], bT = bv[k], bT1 = bv[k + 1], bN1 = bnv[k];
      const double oT = (double)orow[t - t0], oT1 = (double)orow[t - t0 + 1], oN1 = (double)orow[OS + t - t0 + 1];
      const double aJ = (t == 1 && !((double)s.aEntry > LSMALL)) ? LZERO : xp + oT;
      // Every count of a state at a frame is bounded by the state's occupation there (a transition count is a part of it), and that is
      // nothing (below e^-100) for all but the few states along the alignment: the exponentials run only where it is not, and not at all
      // in a wavefront none of whose states is occupied in this frame.
      const double xo = aJ + bT - pr;
#ifdef STATS_EXP_NOOCC
      const bool occd = false;
#else
      const bool occd = inBeam && xo > EXPFLOOR;
#endif
      if (__any(occd)) {
      if (occd) {
         // SetOcct + UpTranParms for state j, and for the entry state at the model's first lane
         double x = xo;
         occJ += (double)(float)exp_sel<FAST>(x, etab);
         if (s.first) {
            x = aE + entry_beta<FAST>(s.aEntry, oT, bT) - pr;
            occE += (double)((x > EXPFLOOR) ? (float)exp_sel<FAST>(x, etab) : 0.0f);
         }
         if (wantTrans) {
            if (s.first) { x = aE + (double)s.aEntry + oT + bT - pr; EXP_TERM(taEntry, x); }
            if (bqt1ok) {
               x = aJ + (double)s.aSelf + (oT1 + bT1) - pr;
               EXP_TERM(taSelf, x);
               if (!s.last) { x = aJ + (double)s.aNext + (oN1 + bN1) - pr; EXP_TERM(taOut, x); }
            }
            if (s.last) {
               // beta_N(q,t): 0 for the last model at T, else beta_1(q+1,t+1) where that is in the beam of t+1
               const double bN = (t == T) ? ((q == Q) ? 0.0 : LZERO)
                                          : ((hasNext && q + 1 >= lo2 && q + 1 <= hi2) ? entry_beta<FAST>(s.aEntryNext, oN1, bN1) : LZERO);
               x = aJ + (double)s.aExit + bN - pr;
               EXP_TERM(taOut, x);
            }
         }
      }
      }
      {
         // UpMixParms seed (HFB.c:1479-1489,1573-1606): the pairs the MINFORPROB prune lets through go to the list of k_mixhits
         double seed = LZERO;
         if (inBeam && wantMix) {
            if (cM == 1 || single) {
               const double x = aJ + bT - pr;
               if (-x < minF) seed = x;
            } else {
               const double initx = xp + (bT - pr);
               const double ub = initx + oT;
               if (ub > -minF - 0.01) seed = initx;
            }
         }
#ifdef STATS_EXP_NOHIT
         const bool hit = false;
#else
         const bool hit = seed > LSMALL;
#endif
         const unsigned long long hm = __ballot(hit);
         if (hm) {
            const int np = __popcll(hm);
            if (hb + np > HB) {                          // the buffer is full: out with it
               for (int i = lane; i < hb; i += 64) { MixHit h; h.st = hbSt[wv][i]; h.frame = hbFr[wv][i]; h.seed = hbSeed[wv][i]; hreg[hc + i] = h; }
               hc += hb; hb = 0;
            }
            if (hit) { const int pos = hb + __popcll(hm & ((1ull << lane) - 1)); hbSt[wv][pos] = a.hitSlots ? ud.slot0 + gl : sidx; hbFr[wv][pos] = ud.frame0 + t - 1; hbSeed[wv][pos] = seed; }
            hb += np;
         }
      }
      }
   }
   for (int i = lane; i < hb; i += 64) { MixHit h; h.st = hbSt[wv][i]; h.frame = hbFr[wv][i]; h.seed = hbSeed[wv][i]; hreg[hc + i] = h; }
   hc += hb;
   if (lane == 0) a.hitCtl[region] = hc;

   // ---- this workgroup's counts.  One transition matrix for the whole chain (a tied-transition system): a row of partial sums per
   // wavefront for k_trans_reduce -- [0..2] a_ii, [3..5] a_i,i+1 (or a_iN from the last state), i = 2..4; [6] a_12; [7..9] occupation
   // of states 2..4; [10] of the entry state; [15] the matrix.  Otherwise atomics per lane, on the lane's own matrix.
   const int cTrans = valid ? a.mTrans[s.mi] : -1;
   const int t0m = a.mTrans[ud.q0];
   const bool uniform = __all(!valid || cTrans == t0m);          // per wavefront: each writes its own row
   if (uniform) {
      double row[11];
#pragma unroll
      for (int i = 2; i <= 4; i++) {
         const bool mine = valid && s.j == i;
         row[i - 2] = mine ? taSelf : 0.0; row[3 + i - 2] = mine ? taOut : 0.0; row[7 + i - 2] = mine ? occJ : 0.0;
      }
      row[6] = taEntry; row[10] = (valid && s.first) ? occE : 0.0;
#pragma unroll
      for (int k = 0; k < 11; k++) {
         double v = row[k];
#pragma unroll
         for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
         row[k] = v;
      }
      if (lane == 0) {
#pragma unroll
         for (int k = 0; k < 11; k++) part[k] = wantTrans ? row[k] : 0.0;
         part[TR_ROW - 1] = wantTrans ? (double)t0m : -1.0;
      }
   } else {
      if (lane == 0) part[TR_ROW - 1] = -1.0;
      if (valid && wantTrans) {
         const int N = s.N, j = s.j;
         double *tr = a.acc + a.lay.tr + a.transOff[cTrans];
         double *oc = a.acc + a.lay.trOcc + a.trOccOff[cTrans];
         if (taSelf != 0.0) atomicAdd(tr + (size_t)(j - 1) * N + (j - 1), taSelf);
         if (taOut != 0.0) atomicAdd(tr + (size_t)(j - 1) * N + j, taOut);
         if (taEntry != 0.0) atomicAdd(tr + 1, taEntry);
         if (occJ != 0.0) atomicAdd(oc + (j - 1), occJ);
         if (s.first && occE != 0.0) atomicAdd(oc, occE);
      }
   }
}

// rows of k_stats_lr -> accumulators: a block sums 256 rows; where they all belong to one matrix (the usual case) one atomic per
// entry and block, else one per entry and row
__global__ __launch_bounds__(256) void k_trans_reduce(FbArgs a, int nRows)
{
   __shared__ double sh[256][TR_ROW + 1];
   __shared__ int mixed;
   const int tid = threadIdx.x, r = blockIdx.x * 256 + tid;
   if (tid == 0) mixed = 0;
   const double *row = a.trPart + (size_t)r * TR_ROW;
   int ti = -1;
   if (r < nRows) ti = (int)row[TR_ROW - 1];
#pragma unroll
   for (int k = 0; k < 11; k++) sh[tid][k] = (ti >= 0) ? row[k] : 0.0;
   __syncthreads();
   // the block's matrix: that of its first live row
   __shared__ int tiBlock;
   if (tid == 0) tiBlock = -1;
   __syncthreads();
   if (ti >= 0) atomicMax(&tiBlock, ti);
   __syncthreads();
   const int tb = tiBlock;
   if (ti >= 0 && ti != tb) mixed = 1;
   __syncthreads();
   auto add_row = [&](int t_, const double *v) {
      const int N0 = a.trOccOff[t_ + 1] - a.trOccOff[t_];
      double *tr = a.acc + a.lay.tr + a.transOff[t_];
      double *oc = a.acc + a.lay.trOcc + a.trOccOff[t_];
      for (int i = 2; i <= 4 && i <= N0 - 1; i++) {
         if (v[i - 2] != 0.0) atomicAdd(tr + (size_t)(i - 1) * N0 + (i - 1), v[i - 2]);
         if (v[3 + i - 2] != 0.0) atomicAdd(tr + (size_t)(i - 1) * N0 + i, v[3 + i - 2]);
         if (v[7 + i - 2] != 0.0) atomicAdd(oc + (i - 1), v[7 + i - 2]);
      }
      if (v[6] != 0.0) atomicAdd(tr + 1, v[6]);
      if (v[10] != 0.0) atomicAdd(oc, v[10]);
   };
   if (tb < 0) return;
   if (mixed) {
      if (ti >= 0) add_row(ti, sh[tid]);
      return;
   }
   // entry k summed over the block's rows by 16 threads each, then over those
   const int k = tid & 15, part = tid >> 4;
   double v = 0.0;
   if (k < 11) for (int rr = part; rr < 256; rr += 16) v += sh[rr][k];
   __syncthreads();
   sh[part][k] = v;
   __syncthreads();
   if (tid < 11) {
      double tot = 0.0;
      for (int p = 0; p < 16; p++) tot += sh[p][tid];
      sh[16][tid] = tot;
   }
   __syncthreads();
   if (tid == 0) add_row(tb, sh[16]);
}

#include "fb_lr_lean.inc"

template <bool FAST> static void launch_beta_lr(const FbArgs &a, int W, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_beta_lr<1, FAST>), dim3(a.nList), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_beta_lr<2, FAST>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_beta_lr<4, FAST>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_beta_lr<8, FAST>), dim3(a.nList), dim3(512), 0, s, a);
}
template <bool FAST> static void launch_alpha_lr(const FbArgs &a, int W, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_alpha_lr<1, FAST>), dim3(a.nList), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_alpha_lr<2, FAST>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_alpha_lr<4, FAST>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_alpha_lr<8, FAST>), dim3(a.nList), dim3(512), 0, s, a);
}
template <bool FAST> static void launch_stats_lr(const FbArgs &a, int W, int nChunks, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_stats_lr<1, FAST>), dim3(a.nList, nChunks), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_stats_lr<2, FAST>), dim3(a.nList, nChunks), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_stats_lr<4, FAST>), dim3(a.nList, nChunks), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_stats_lr<8, FAST>), dim3(a.nList, nChunks), dim3(512), 0, s, a);
}

int htkamd_stats_lr_chunks(int TMax) { return (TMax + STATS_FC - 1) / STATS_FC; }
int htkamd_stats_lr_region_cap(void) { return STATS_FC * 64; }
size_t htkamd_stats_lr_row_doubles(void) { return TR_ROW; }

// experiment / fallback switch: HTKAMD_LR_LEAN=0 keeps the round-4 kernels (bit 0: beta, bit 1: alpha, bit 2: statistics, bit 3: one
// wavefront with two states per lane for chains of 65 .. 128 states; default 15)
static int lr_lean_mask()
{
   static const int m = [] { const char *e = getenv("HTKAMD_LR_LEAN"); return e ? atoi(e) : 15; }();
   return m;
}
#define LAUNCH_W(K, ...) \
   do { \
      if (W == 1) hipLaunchKernelGGL((K<1, ##__VA_ARGS__>), grid, dim3(64), 0, s, a); \
      else if (W == 2) hipLaunchKernelGGL((K<2, ##__VA_ARGS__>), grid, dim3(128), 0, s, a); \
      else if (W == 4) hipLaunchKernelGGL((K<4, ##__VA_ARGS__>), grid, dim3(256), 0, s, a); \
      else hipLaunchKernelGGL((K<8, ##__VA_ARGS__>), grid, dim3(512), 0, s, a); \
   } while (0)

bool htkamd_stats_lr_is_sparse(const FbArgs &a) { return a.laneRec && (lr_lean_mask() & 4); }
// no pruning beam and the fp32-transcendental class: the lean kernels (fb_lr_lean.inc)
bool htkamd_beta_lr_is_lean(const FbArgs &a, bool fast) { return fast && a.qBeamNP && !(a.pruneInit < HTKAMD_NOPRUNE) && (lr_lean_mask() & 1); }

int htkamd_launch_beta_lr(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (htkamd_beta_lr_is_lean(a, fast)) {
      const dim3 grid(a.nList);
      if (W == 2 && (lr_lean_mask() & 8)) hipLaunchKernelGGL(k_beta_np2, grid, dim3(64), 0, s, a);      // one wavefront, two states per lane
      else if (W == 1) hipLaunchKernelGGL((k_beta_np<1>), grid, dim3(64), 0, s, a);
      else if (W == 2) hipLaunchKernelGGL((k_beta_np<2>), grid, dim3(128), 0, s, a);
      else if (W == 4) hipLaunchKernelGGL((k_beta_np<4>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_beta_np<8>), grid, dim3(512), 0, s, a);
      HIPCHECK(hipGetLastError());
      return HTKAMD_OK;
   }
   if (fast) launch_beta_lr<true>(a, W, s); else launch_beta_lr<false>(a, W, s);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_launch_alpha_lr(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (fast && (lr_lean_mask() & 2)) {
      const dim3 grid(a.nList);
      if (W == 2 && (lr_lean_mask() & 8)) {
         if (a.alphaDbg) hipLaunchKernelGGL(k_alpha_f2<true>, grid, dim3(64), 0, s, a); else hipLaunchKernelGGL(k_alpha_f2<false>, grid, dim3(64), 0, s, a);
      }
      else if (a.alphaDbg) LAUNCH_W(k_alpha_f, true); else LAUNCH_W(k_alpha_f, false);
      HIPCHECK(hipGetLastError());
      return HTKAMD_OK;
   }
   if (fast) launch_alpha_lr<true>(a, W, s); else launch_alpha_lr<false>(a, W, s);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// the frame-parallel statistics of the utterances whose alpha columns are stored, and the reduction of their partial rows (a.trPart,
// a.hits, a.hitCtl: room for nList * chunks * W rows / regions)
int htkamd_launch_stats_lr(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   const int nChunks = htkamd_stats_lr_chunks(a.TMax);
   if (htkamd_stats_lr_is_sparse(a)) {
      const dim3 grid(a.nList, nChunks);
      if (fast) LAUNCH_W(k_stats_sp, true); else LAUNCH_W(k_stats_sp, false);
   }
   else if (fast) launch_stats_lr<true>(a, W, nChunks, s); else launch_stats_lr<false>(a, W, nChunks, s);
   const int nRows = a.nList * nChunks * W;
   hipLaunchKernelGGL(k_trans_reduce, dim3((nRows + 255) / 256), dim3(256), 0, s, a, nRows);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}
