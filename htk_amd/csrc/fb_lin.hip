// fb_lin.hip -- K2l / K3l: the forward-backward recursions in the SCALED LINEAR domain (scoreMode bit HTKAMD_SCORE_LINEAR).
//
// Same mapping as fb_state.hip (one lane per emitting chain state, W wavefronts per utterance, ONE LDS exchange per frame), same
// reference semantics (SetBeta HFB.c:1149, StepAlpha :686, MaxModelProb :655, SetOcct :399, UpTranParms :1371, UpMixParms seeds
// :1479), tolerance class.  What changes is the arithmetic of a step.  The reference keeps alpha and beta as logarithms and pays an
// exp and a log for every term of every sum (LAdd, HMath.c:1576); here a column holds probabilities divided by a per-frame scale,
//      beta_j(t)  = c_j(t) * exp(SB(t)),        alpha_j(t) = ahat_j(t) * exp(SA(t)),
// the sums are fp64 multiply-adds, and only one exponential per state and frame remains: the emission factor
//      bhat_j(t) = exp(outp_j(t) - omax_t),     omax_t = the largest score of frame t among the states the pass may touch.
// Scales.  A column computed from the column before is divided by 2^X, X = the largest binary exponent in THAT column (an exact
// scaling: v_ldexp_f64), so a column's magnitude depends on one frame's emission factors only -- there is no feedback and no drift:
//      c(t)    = 2^-X(t+1) * sum a_jk p_k(t+1),   p_k(t) = bhat_k(t) c_k(t),   SB(t) = SB(t+1) + omax_(t+1) + X(t+1) ln 2,  SB(T) = 0
//      ahat(t) = bhat(t) * u(t),  u_j(t) = 2^-XA(t-1) * (a_1j alpha_1 + sum a_ij ahat_i(t-1)),  SA(t) = SA(t-1) + XA(t-1) ln 2 + omax_t
// X comes from a 6-step DPP maximum of the lanes' exponents inside each wavefront and rides on the step's LDS exchange between
// wavefronts; omax_t is reduced the same way one frame ahead.  Values more than ~e^-700 below their column's largest underflow to
// zero where the reference keeps a logarithm; their share of any statistic is below anything a float accumulator registers.
// Everything the statistics need is a product with one of two per-frame constants,
//      kappa_t = exp(SA(t) + SB(t) - pr)          occupation counts, entry transitions, mixture seeds, MaxModelProb
//      tau_t   = kappa_t * 2^-X(t+1)              transitions into column t+1 (out, exit), the previous model's exit term
// so the arithmetic of a step shrinks (no LAdd, no per-term exp).  Accuracy: fp64 products and sums (1e-16 per operation), emission
// factors and kappa from v_exp_f32 on a reduced argument (1e-7 relative): utterance log-probabilities agree with the reference to
// ~1e-9 relative, accumulators well inside the 1e-4 bar (tests/test_gpu_parity.py).
// MEASURED (round 2, 5k x 16, 1250 x 500 frames): NOT faster than the log-domain kernels with the fp32-transcendental LAdd -- beta
// 1.29 ms against 0.74 ms, alpha 2.21 against 1.52.  The recursions are bound by instruction issue on 2-3 wavefronts per SIMD (about
// 120 vector + 115 scalar instructions per wavefront and frame in fb_state.hip); the two wavefront reductions per frame that the
// scaling needs (exponent, omax: ~20 vector instructions each with their DPP wait states; without them beta runs in 0.94 ms), the
// conversions (exp2_split, ldexp, frexp) and the extra per-frame record cost what the LAdds saved.  Kept as a selectable mode and
// as the reference point for that experiment; bench.py and the drivers use HTKAMD_SCORE_FASTLADD.
//
// Beta goes out as in fb_state.hip (betaS[t][lane] = c_j(t), betaE[t][first lane of q] = sum_k a_1k p_k(t), the latter in the scale
// SB(t) + omax_t) plus one LinFrame {SB(t), omax_t, X(t)} per frame.
#include <hip/hip_runtime.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "wavegrp.h"
#include "fb_state.h"

#define LOG2E 1.4426950408889634
#define LN2 0.6931471805599453
#define NOEXP (-100000)

// ---- wavefront maxima through DPP row shifts / broadcasts (max is idempotent: no masks needed); the result is in lane 63
__device__ __forceinline__ int wave_max_i32(int v)
{
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));      // row_shr:1
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));      // row_shr:2
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));      // row_shr:4
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));      // row_shr:8
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xf, 0xf, false));      // row_bcast:15
   v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xf, 0xf, false));      // row_bcast:31
   return __builtin_amdgcn_readlane(v, 63);
}
// floats compare like sign-magnitude integers: map to a monotone int key, reduce, map back
__device__ __forceinline__ float wave_max_f32(float x)
{
   int k = __float_as_int(x);
   k = (k >= 0) ? k : (int)(0x80000000u - (unsigned)k);
   k = wave_max_i32(k);
   k = (k >= 0) ? k : (int)(0x80000000u - (unsigned)k);
   return __int_as_float(k);
}

// 2^z for a double z, relative error ~1e-7 (v_exp_f32 on the fraction), exact scaling by the integer part; 0 for very negative z
__device__ __forceinline__ double exp2_split(double z)
{
   const double zi = rint(z);
   const float f = (float)(z - zi);
   const double m = (double)__builtin_amdgcn_exp2f(f);
   return ldexp(m, (int)zi);
}
// log(v) for a double v > 0, absolute error ~1e-7
__device__ __forceinline__ double log_split(double v)
{
   const int e = __builtin_amdgcn_frexp_exp(v);
   const float m = (float)__builtin_amdgcn_frexp_mant(v);
   return ((double)e + (double)__builtin_amdgcn_logf(m)) * LN2;
}
__device__ __forceinline__ double lin_of(float a) { return (a > (float)LSMALL) ? exp((double)a) : 0.0; }
__device__ __forceinline__ int exp_of(double v) { return (v > 0.0) ? __builtin_amdgcn_frexp_exp(v) : NOEXP; }

#define LIN_AT(t) (a.lin[ud.frame0 + (t) - 1])

// ------------------------------------------------------------------------------------ K2l: beta
template <int W>
__global__ __launch_bounds__(64 * W) void k_beta_l(FbArgs a)
{
   constexpr int L = 64 * W, LP = L + 2 * SPAD;
   __shared__ double xp[2][LP];                        // p_j = bhat_j c_j of the column just computed, by step parity
   __shared__ double xc[2][LP];                        // c_j of that column (beam pruning only)
   __shared__ int xe[2][8];                            // per wavefront: largest exponent of p in the column
   __shared__ float xo[2][8];                          // per wavefront: largest score of the NEXT frame to be worked on
   __shared__ float stage[W][2 * 64 * 4];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   __shared__ short sqOf[L];
   __shared__ short flOf[L + 2];
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
   StateRegs s;
   load_state(s, a, ud, gl, valid);
   float aExitPrev[3], aEntryNext[3];
   load_neighbours(aExitPrev, aEntryNext, a, ud, s, valid);
   for (int i = gl; i < LP; i += L) { xp[0][i] = 0.0; xp[1][i] = 0.0; xc[0][i] = 0.0; xc[1][i] = 0.0; }
   sqOf[gl] = (short)(valid ? s.q : Q + 1);
   if (valid && s.first) flOf[s.q] = (short)gl;
   if (gl == 0) { flOf[Q + 1] = (short)nS; flOf[0] = 0; }
   __syncthreads();
   bool useOut[5], useEnt[3];
#pragma unroll
   for (int d = 0; d < 5; d++) useOut[d] = g.ballot(valid && s.aOut[d] > (float)LSMALL).highest() >= 0;
#pragma unroll
   for (int k = 0; k < 3; k++) useEnt[k] = g.ballot(valid && ((s.first && s.aEntryOf[k] > (float)LSMALL) || aEntryNext[k] > (float)LSMALL)).highest() >= 0;
   double lOut[5], lEntNext[3], lEntOf[3];
#pragma unroll
   for (int d = 0; d < 5; d++) lOut[d] = lin_of(s.aOut[d]);
#pragma unroll
   for (int k = 0; k < 3; k++) { lEntNext[k] = lin_of(aEntryNext[k]); lEntOf[k] = lin_of(s.aEntryOf[k]); }
   const double lExit = lin_of(s.aExit);
   const int q = s.q, N = s.N, j = s.j;
   const int offNext = N - j;
   const short *tLo = a.taperLo + ud.frame0 - 1, *tHi = a.taperHi + ud.frame0 - 1;   // 1-based t
   short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
   ObsRow st;
   st.lds = stage[wv]; st.lane = lane; st.R = (f4s)(0.f);
   st.row = valid ? a.outp + ud.outp0 + (size_t)gl * T : nullptr;
   const bool pruning = a.pruneInit < HTKAMD_NOPRUNE;

   // everything a step needs of the column before it, read in ONE pass after the single exchange of a step
#define LIN_GATHER(par, wantOwn)                                                                                                  \
   do {                                                                                                                           \
      const double *xp_ = xp[par] + SPAD + gl;                                                                                     \
      _Pragma("unroll") for (int d = 0; d < 5; d++) if (useOut[d]) pS[d] = xp_[d - 2];                                             \
      double x_ = 0.0;                                                                                                            \
      _Pragma("unroll") for (int k = 0; k < 3; k++) if (useEnt[k]) x_ += lEntNext[k] * xp_[offNext + k];                           \
      bEn = x_;                                                                                                                   \
      lMax = 0.0;                                                                                                                 \
      if ((wantOwn) && s.first) {                                                                                                 \
         x_ = 0.0;                                                                                                                \
         _Pragma("unroll") for (int k = 0; k < 3; k++) if (2 + k <= N - 1) x_ += lEntOf[k] * xp_[k];                               \
         bE = x_;                                                                                                                 \
         if (pruning) {                                                                                                           \
            const double *xc_ = xc[par] + SPAD + gl;                                                                               \
            _Pragma("unroll") for (int k = 0; k < 3; k++) if (2 + k <= N - 1) { const double y = xc_[k]; if (y > lMax) lMax = y; } \
         }                                                                                                                        \
      }                                                                                                                           \
      int X_ = xe[par][0]; float o_ = xo[par][0];                                                                                 \
      _Pragma("unroll") for (int w = 1; w < W; w++) { X_ = max(X_, xe[par][w]); o_ = fmaxf(o_, xo[par][w]); }                      \
      Xn = (X_ > NOEXP / 2) ? X_ : 0; omaxNext = o_;                                                                               \
   } while (0)

   double thresh = a.pruneInit, pr = LZERO;
   int ok = 0;
   for (;;) {                                            // StepBack retry loop (HFB.c:1332-1361)
      int fail = 0;
      const double eThr = pruning ? exp(-thresh) : 0.0;
      double cJ = 0.0, bE = 0.0, bEn = 0.0, lMax = 0.0;
      double pS[5];
      float obT = 0.f, obP = 0.f;
#pragma unroll
      for (int d = 0; d < 5; d++) pS[d] = 0.0;
      int Xn = 0; float omaxNext = 0.f;
      double SB = 0.0;                                   // scale of the column in cJ
      // ---- t = T (HFB.c:1175-1198)
      const int endT = tLo[T];
      {
         const int bl = (T - 1) >> 2;
         st.load(bl); st.park(bl);
         if (bl >= 1) st.load(bl - 1);
         obT = st.get(T - 1);
         if (T >= 2) {
            if (((T - 2) & 3) == 3) { st.park((T - 2) >> 2); if (((T - 2) >> 2) >= 1) st.load(((T - 2) >> 2) - 1); }
            obP = st.get(T - 2);
         }
      }
      // the models that have a score at frame t in any pass (fb.hip evLo/evHi): what omax_t is taken over
      int uHiN = Q, uLoN = endT;
      float omaxT;
      {
         const int e0 = (endT > 1) ? endT - 1 : 1;
         const float m = wave_max_f32((valid && q >= e0) ? obT : -3.0e38f);
         if (lane == 0) xo[0][wv] = m;
         xsync<W>();
         float o_ = xo[0][0];
#pragma unroll
         for (int w = 1; w < W; w++) o_ = fmaxf(o_, xo[0][w]);
         omaxT = o_;
         xsync<W>();
      }
      const bool inT = valid && q >= endT;
      float omaxPub = 0.f;                               // omax of the frame below the one being worked on, to be published
      {
         // exit value at T: 1 for the last model, 0 for the others (no tee models in this path)
         cJ = (inT && q == Q) ? lExit : 0.0;
         const double p = inT ? exp2_split(((double)obT - (double)omaxT) * LOG2E) * cJ : 0.0;
         const int we = wave_max_i32(exp_of(p));
         if (T >= 2) {
            const int startq = uHiN, endq = (uLoN == 1) ? 1 : ((tLo[T - 1] >= uLoN) ? tLo[T - 1] : uLoN - 1);
            const int e0 = (endq > 1) ? endq - 1 : 1;
            omaxPub = wave_max_f32((valid && q >= e0 && q <= startq) ? obP : -3.0e38f);
            uHiN = (tHi[T - 1] < startq) ? tHi[T - 1] : startq; uLoN = endq;
         }
         xp[T & 1][SPAD + gl] = p; if (pruning) xc[T & 1][SPAD + gl] = cJ;
         if (lane == 0) { xe[T & 1][wv] = we; xo[T & 1][wv] = omaxPub; }
         xsync<W>();
         LIN_GATHER(T & 1, inT);
         if (inT) { BETA_S(T) = cJ; if (s.first) BETA_E(T) = bE; }
      }
      if (gl == 0) { gLo[T] = (short)endT; gHi[T] = (short)Q; LinFrame lf; lf.sb = 0.0; lf.omax = omaxT; lf.xb = Xn; LIN_AT(T) = lf; }
      int qHiN = Q, qLoN = endT, lastEnd = endT;
      int nxtLo = (T >= 2) ? tLo[T - 1] : 1, nxtHi = (T >= 2) ? tHi[T - 1] : 1;
      bool stPrev = false, stIn = false; int tPrev = 0, loPrev = 1, hiPrev = 1;
      double SBprev = 0.0; float omaxPrev = 0.f;
      double bE1 = bE; double SB1 = 0.0; float omax1 = omaxT;   // of the last column worked on (for pr)

      // ---- t = T-1 .. 1 (HFB.c:1205-1277)
      for (int t = T - 1; t >= 1; t--) {
         const int taperLoT = nxtLo, taperHiT = nxtHi;
         if (t >= 2) { nxtLo = tLo[t - 1]; nxtHi = tHi[t - 1]; }
         if (t >= 2 && ((t - 2) & 3) == 3) { st.park((t - 2) >> 2); if (((t - 2) >> 2) >= 1) st.load(((t - 2) >> 2) - 1); }
         if (stPrev) {                                   // the column finished in the previous iteration goes out now
            if (stIn) { BETA_S(tPrev) = cJ; if (s.first) BETA_E(tPrev) = bE; }
            if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; LinFrame lf; lf.sb = SBprev; lf.omax = omaxPrev; lf.xb = Xn; LIN_AT(tPrev) = lf; }
            stPrev = false;
         }
         // scale of the column of this step: the column before was divided by nothing, its p-scale is SB + omax, and this step divides by 2^Xn
         SB = SB + (double)omaxT + (double)Xn * LN2;
         omaxT = omaxNext;                               // omax of frame t, published during the step before
         obT = obP;
         if (t >= 2) obP = st.get(t - 2);
         // omax of frame t-1, one step ahead
         if (t >= 2) {
            const int startq = uHiN, endq = (uLoN == 1) ? 1 : ((nxtLo >= uLoN) ? nxtLo : uLoN - 1);
            const int e0 = (endq > 1) ? endq - 1 : 1;
            omaxPub = wave_max_f32((valid && q >= e0 && q <= startq) ? obP : -3.0e38f);
            uHiN = (nxtHi < startq) ? nxtHi : startq; uLoN = endq;
         }
         const int startq = qHiN;
         const int endq = (qLoN == 1) ? 1 : ((taperLoT >= qLoN) ? taperLoT : qLoN - 1);
         const bool inRange = valid && q >= endq && q <= startq;
         const bool wasIn = q >= qLoN && q <= qHiN;
         double p = 0.0;
         if (inRange) {
            const bool p1 = (q < Q) && (q + 1 >= qLoN) && (q + 1 <= qHiN);
            double x = p1 ? lExit * bEn : 0.0;                   // beta_N(q,t) = beta_1(q+1,t+1)
            if (wasIn) {
#pragma unroll
               for (int d = 0; d < 5; d++) if (useOut[d]) x += lOut[d] * pS[d];
            }
            cJ = ldexp(x, -Xn);
            p = exp2_split(((double)obT - (double)omaxT) * LOG2E) * cJ;
         }
         const int we = wave_max_i32(exp_of(p));
         xp[t & 1][SPAD + gl] = p; if (pruning) xc[t & 1][SPAD + gl] = inRange ? cJ : 0.0;
         if (lane == 0) { xe[t & 1][wv] = we; xo[t & 1][wv] = omaxPub; }
         xsync<W>();
         LIN_GATHER(t & 1, inRange);
         int newHi, newLo;
         if (!pruning) {                                 // only the taper acts (HFB.c:1259-1264)
            newHi = (taperHiT < startq) ? taperHiT : startq;
            newLo = endq;
         } else {                                        // beam pruning (HFB.c:1254-1272): one bit per model, at its first lane
            const bool rep = inRange && s.first;
            const double gmax = g.maxall(rep ? lMax : 0.0);
            const bool drop = gmax > 0.0 && (lMax <= 0.0 || lMax < gmax * eThr);     // gmax - lMax > thresh
            const MaskW<W> keep = g.ballot(rep && !drop);
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
         stPrev = true; stIn = inRange; tPrev = t; loPrev = newLo; hiPrev = newHi; SBprev = SB; omaxPrev = omaxT;
         qHiN = newHi; qLoN = newLo; lastEnd = endq;
         bE1 = bE; SB1 = SB; omax1 = omaxT;
      }
      if (!fail && stPrev) {                             // the last column (t = 1)
         if (stIn) { BETA_S(tPrev) = cJ; if (s.first) BETA_E(tPrev) = bE; }
         if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; LinFrame lf; lf.sb = SBprev; lf.omax = omaxPrev; lf.xb = Xn; LIN_AT(tPrev) = lf; }
      }
      if (!fail) {
         const double b1 = g.bcast(bE1, flOf[lastEnd]);  // utt->pr = beta_1 of the last model processed
         pr = (b1 > 0.0) ? log(b1) + SB1 + (double)omax1 : LZERO;
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
#undef LIN_GATHER
}

// ------------------------------------------------------------------------------------ K3l: alpha + stats
template <int W>
__global__ __launch_bounds__(64 * W) void k_alpha_l(FbArgs a)
{
   constexpr int L = 64 * W, LP = L + 2 * SPAD;
   __shared__ double xalpha[2][LP];                    // ahat_j(t) by step parity
   __shared__ double xsum[2][LP];                      // ahat_j(t) c_j(t) inside the beta beam (MaxModelProb)
   __shared__ double xnext[2][LP];                     // p_j(t+1) inside the beam of t+1 (transition counts)
   __shared__ int xe[2][8];
   __shared__ float stage[W][2 * 64 * 4];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   __shared__ short sqOf[L];
   __shared__ short flOf[L + 2];
   const int gl = threadIdx.x, lane = gl & 63, wv = gl >> 6;
   const int li = blockIdx.x;
   if (li >= a.nList) return;
   const int u = a.uttList[li];
   const UttDesc ud = a.utt[u];
   if (a.status[u] != HTKAMD_UTT_OK) {                   // skipped in the beta pass (or pre-check)
      if (gl == 0) atomicAdd(a.acc + a.lay.nUttSkipped, 1.0);
      return;
   }
   Grp<W> g; g.x = gx; g.a1 = ga1; g.wave = wv; g.lane = lane; g.ph = 0;
   const int T = ud.T, Q = ud.Q, nS = ud.nSlots, nC = ud.nCells;
   const bool valid = gl < nS;
   StateRegs s;
   load_state(s, a, ud, gl, valid);
   float aExitPrev[3], aEntryNext[3];
   load_neighbours(aExitPrev, aEntryNext, a, ud, s, valid);
   for (int i = gl; i < LP; i += L) {
#pragma unroll
      for (int k = 0; k < 2; k++) { xalpha[k][i] = 0.0; xsum[k][i] = 0.0; xnext[k][i] = 0.0; }
   }
   sqOf[gl] = (short)(valid ? s.q : Q + 1);
   if (valid && s.first) flOf[s.q] = (short)gl;
   if (gl == 0) { flOf[Q + 1] = (short)nS; flOf[0] = 0; }
   __syncthreads();
   bool useIn[5], useExitP[3], useOut[5];
#pragma unroll
   for (int d = 0; d < 5; d++) { useIn[d] = g.ballot(valid && s.aIn[d] > (float)LSMALL).highest() >= 0; useOut[d] = g.ballot(valid && s.aOut[d] > (float)LSMALL).highest() >= 0; }
#pragma unroll
   for (int k = 0; k < 3; k++) useExitP[k] = g.ballot(valid && (aExitPrev[k] > (float)LSMALL || (s.last && s.aExitOf[k] > (float)LSMALL))).highest() >= 0;
   const MaskW<W> firsts = g.ballot(valid && s.first);
   double lIn[5], lOut[5], lExitPrev[3], lExitOf[3];
#pragma unroll
   for (int d = 0; d < 5; d++) { lIn[d] = lin_of(s.aIn[d]); lOut[d] = lin_of(s.aOut[d]); }
#pragma unroll
   for (int k = 0; k < 3; k++) { lExitPrev[k] = lin_of(aExitPrev[k]); lExitOf[k] = lin_of(s.aExitOf[k]); }
   const double lEntry = lin_of(s.aEntry), lExit = lin_of(s.aExit);
   const int q = s.q, N = s.N, j = s.j;
   const int offNext = N - j, offPrev = -(j - 1);
   const int cHmm = valid ? a.mHmm[s.mi] : 0, cTrans = valid ? a.mTrans[s.mi] : 0, mc0 = valid ? a.mCell0[s.mi] : 0;
   int cM = 0;
   if (valid) { const int sidx = a.slotState[ud.slot0 + gl]; cM = a.stateCompOff[sidx + 1] - a.stateCompOff[sidx]; }
   const short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
   short *gaLo = a.aLo + ud.frame0 - 1, *gaHi = a.aHi + ud.frame0 - 1;
   ObsRow st;
   st.lds = stage[wv]; st.lane = lane; st.R = (f4s)(0.f);
   st.row = valid ? a.outp + ud.outp0 + (size_t)gl * T : nullptr;
   double *gam = a.gam + ud.gam0 + gl;
   const double *bNextE = a.betaW + ud.betaW0 + (size_t)T * L + (gl + offNext);
   const bool hasNext = valid && q < Q;
   const double pr = a.pr[u];
   const double minF = (double)a.minFrwdP;
   const double eMinF = exp(-minF), eMinF2 = exp(-minF - 0.01);
   const bool wantMix = (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES)) != 0;
   const bool wantTrans = (a.uFlags & HTKAMD_UPTRANS) != 0;
   const bool oneMix = (cM == 1 || a.maxM == 1);

   double aJ = 0.0, aEh = 0.0, aEnext = 0.0;             // ahat_j(t); alpha_1(q,t) in the scale of u; exit value of the model before, column t
   double yIn[5];
   double uPre = 0.0;
   double taOut[5], taExit = 0.0, taEntry = 0.0, occJ = 0.0, occE = 0.0;
#pragma unroll
   for (int d = 0; d < 5; d++) { taOut[d] = 0.0; yIn[d] = 0.0; }
   double bT = 0.0, bT1 = 0.0, bT2 = 0.0, eT = 0.0, eT1 = 0.0, eT2 = 0.0, nT1 = 0.0, nT2 = 0.0;
   float oT = 0.f, oT1 = 0.f;
   int lo0 = 1, hi0 = 0, lo1 = gLo[1], hi1 = gHi[1], lo2 = (T >= 2) ? gLo[2] : 1, hi2 = (T >= 2) ? gHi[2] : 0, lo3 = 1, hi3 = 0;
   LinFrame lf0 = LIN_AT(1), lf1 = lf0, lf2 = lf0;
   if (T >= 2) lf1 = LIN_AT(2);
   if (valid) {
      bT = BETA_S(1); if (s.first) eT = BETA_E(1);
      if (T >= 2) { bT1 = BETA_S(2); if (s.first) eT1 = BETA_E(2); if (hasNext) nT1 = bNextE[(size_t)1 * L]; }
   }
   st.load(0); st.park(0);
   if (T > 4) st.load(1);
   oT = st.get(0); if (T >= 2) oT1 = st.get(1);
   int sq = 1, eq = hi1, err = 0;
   bool mmDrop = false;                                  // pr - MaxModelProb(q, t-1) > minFrwdP, at the model's first lane
   double SA = 0.0; int XA = 0;                          // scale of the column before, its largest exponent
   double pT = 0.0;                                      // p_j(t) = bhat_j(t) c_j(t)
   double bh = valid ? exp2_split(((double)oT - (double)lf0.omax) * LOG2E) : 0.0;

   for (int t = 1; t <= T; t++) {
      if (t + 2 <= T) {
         lo3 = gLo[t + 2]; hi3 = gHi[t + 2]; lf2 = LIN_AT(t + 2);
         if (valid) { bT2 = BETA_S(t + 2); if (s.first) eT2 = BETA_E(t + 2); if (hasNext) nT2 = bNextE[(size_t)(t + 1) * L]; }
      }
      // emission factor of t+1 (for p_j(t+1)); that of t was computed in the step before
      const double bh1 = (valid && t < T) ? exp2_split(((double)oT1 - (double)lf1.omax) * LOG2E) : 0.0;
      pT = bh * bT;
      const double pT1 = bh1 * bT1;
      bool in;
      if (t == 1) {
         // ---- InitAlpha (HFB.c:616-651): without tee models only the first model starts
         in = valid && q <= eq;
         aEh = (in && q == 1) ? 1.0 : 0.0;
         uPre = lEntry * aEh;
         aJ = in ? bh * uPre : 0.0;
      } else {
         // ---- alpha beam (HFB.c:699-722) from MaxModelProb of column t-1: one bit per model at its first lane
         const MaskW<W> kept = firsts & ~g.ballot(valid && s.first && mmDrop);
         const int slane = (kept & MaskW<W>::range(flOf[lo0], L - 1)).lowest();
         int sN = (slane >= 0) ? sqOf[slane] : Q + 1;
         if (sN < 1 || sN > hi1) { err = 1; break; }
         if (sN < lo1) sN = lo1;
         int e = (hi0 < Q) ? hi0 + 1 : hi0;
         const int elane = (kept & MaskW<W>::range(0, flOf[e + 1] - 1)).highest();
         e = (elane >= 0) ? sqOf[elane] : 0;
         if (e < 1 || e < sN) { err = 1; break; }
         if (e > hi1) e = hi1;
         sq = sN; eq = e;
         // ---- alpha column t (HFB.c:729-771)
         in = valid && q >= sq && q <= eq;
         aJ = 0.0; aEh = 0.0; uPre = 0.0;
         if (in) {
            const double a1 = (q == 1) ? 0.0 : aEnext;               // alpha_1(q,t) = alpha_N(q-1,t-1)
            double x = lEntry * a1;
#pragma unroll
            for (int d = 0; d < 5; d++) if (useIn[d]) x += lIn[d] * yIn[d];
            aEh = ldexp(a1, -XA);
            uPre = ldexp(x, -XA);
            aJ = bh * uPre;
         }
      }
      if (gl == 0) { gaLo[t] = (short)sq; gaHi[t] = (short)eq; }
      // scale of this column
      SA = SA + (double)XA * LN2 + (double)lf0.omax;

      // ---- the one exchange of the step
      const bool inB = valid && q >= lo1 && q <= hi1;    // in the beta beam of t
      const bool inBeam = in;
      const bool bqt1ok = (t < T) && q >= lo2 && q <= hi2;
      const int par = t & 1;
      const int we = wave_max_i32(exp_of(aJ));
      xalpha[par][SPAD + gl] = aJ;
      xsum[par][SPAD + gl] = inB ? aJ * bT : 0.0;
      xnext[par][SPAD + gl] = (valid && bqt1ok) ? pT1 : 0.0;
      if (lane == 0) xe[par][wv] = we;
      xsync<W>();
      {
         int X_ = xe[par][0];
#pragma unroll
         for (int w = 1; w < W; w++) X_ = max(X_, xe[par][w]);
         XA = (X_ > NOEXP / 2) ? X_ : 0;
      }
      // kappa = 2^kE * kM and tau = 2^tE * kM: max_j alpha_j * max_j beta_j may exceed pr by far more than a double's range (the two maxima
      // sit in different states), so the factors are applied as mantissa and exponent -- KAP(v) = v * kappa without ever forming kappa
      const double zK = (SA + lf0.sb - pr) * LOG2E, zKi = rint(zK);
      const double kM = (double)__builtin_amdgcn_exp2f((float)(zK - zKi));
      const int kE = (zKi > 4000.0) ? 4000 : ((zKi < -4000.0) ? -4000 : (int)zKi);
      const int tE = (t < T) ? kE - lf1.xb : kE;                       // at T: beta_N of the last model is 1 and SB(T) = 0
#define KAP(v) ldexp((v) * kM, kE)
#define TAU(v) ldexp((v) * kM, tE)
      const double *xa = xalpha[par] + SPAD + gl;
#pragma unroll
      for (int d = 0; d < 5; d++) if (useIn[d]) yIn[d] = xa[d - 2];
      // exit value of the model BEFORE this one in column t: alpha_1 of this model in column t+1
      double aXp = 0.0;
      if (valid && q > 1) {
#pragma unroll
         for (int k = 2; k >= 0; k--) if (useExitP[k]) aXp += lExitPrev[k] * xa[offPrev - k];
      }
      if (a.alphaDbg && valid) {
         double *ad = a.alphaDbg + ud.beta0 + (size_t)(t - 1) * nC + mc0;
         ad[j - 1] = (aJ > 0.0) ? log(aJ) + SA : LZERO;
         if (s.first) ad[0] = (aEh > 0.0) ? log(aEh) + SA - (double)lf0.omax : LZERO;
         if (s.last) {
            double x = 0.0;
#pragma unroll
            for (int k = 2; k >= 0; k--) if (N - 1 - k >= 2) x += lExitOf[k] * xa[-k];
            ad[N - 1] = (x > 0.0) ? log(x) + SA : LZERO;
         }
      }
      // ---- statistics for column t (HFB.c:1790-1806) and MaxModelProb of column t
      double bN = 0.0;
      if (valid) bN = (t == T) ? ((q == Q) ? 1.0 : 0.0) : ((hasNext && q + 1 >= lo2 && q + 1 <= hi2) ? nT1 : 0.0);
      if (valid && s.first) {
         double mm = 0.0;
         if (inB) {
            mm = aEh * eT;                               // i = 1
            const double *xs = xsum[par] + SPAD + gl;
#pragma unroll
            for (int k = 0; k < 3; k++) if (2 + k <= N - 1) { const double v = xs[k]; if (v > mm) mm = v; }
            mm = KAP(mm);
         }
         double prevExit = 0.0;
         if (q > 1 && q - 1 >= lo1 && q - 1 <= hi1) {
            const double bNp = (t == T) ? 0.0 : ((q >= lo2 && q <= hi2) ? eT1 : 0.0);
            prevExit = TAU(aXp * bNp);
         }
         const double mmp = (prevExit > mm) ? prevExit : mm;
         mmDrop = mmp < eMinF;                           // pr - MaxModelProb > minFrwdP
      }
      if (inBeam) {
         occJ += KAP(aJ * bT);
         if (s.first) occE += KAP(aEh * eT);
         if (wantTrans) {
            taEntry += KAP(aEh * pT * lEntry);
            if (bqt1ok) {
               const double *xn = xnext[par] + SPAD + gl;
#pragma unroll
               for (int d = 0; d < 5; d++) if (useOut[d]) taOut[d] += TAU(aJ * xn[d - 2] * lOut[d]);
            }
            taExit += TAU(aJ * bN * lExit);
         }
      }
      if (valid) {
         // UpMixParms seed (HFB.c:1479-1489,1573-1606)
         double seed = LZERO;
         if (inBeam && wantMix) {
            const double gamma = KAP(aJ * bT);
            if (oneMix) { if (gamma > eMinF) seed = log_split(gamma); }
            else if (gamma > eMinF2) seed = log_split(KAP(uPre * bT)) - (double)lf0.omax;
         }
         gam[(size_t)(t - 1) * nS] = seed;
      }
      aEnext = aXp;
      // rotate: t -> t+1
      bT = bT1; bT1 = bT2; eT = eT1; eT1 = eT2; nT1 = nT2;
      oT = oT1; bh = bh1;
      if (t + 2 <= T) {
         const int f = t + 1;
         if ((f & 3) == 0) { st.park(f >> 2); if (4 * ((f >> 2) + 1) < T) st.load((f >> 2) + 1); }
         oT1 = st.get(f);
      }
      lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2; lo2 = lo3; hi2 = hi3;
      lf0 = lf1; lf1 = lf2;
#undef KAP
#undef TAU
   }

   if (err) {
      if (gl == 0) { a.status[u] = HTKAMD_UTT_EALPHA; atomicAdd(a.acc + a.lay.nUttSkipped, 1.0); }
      return;
   }
   // ---- flush
   // (the transition probabilities are inside the sums: where a transition does not exist the rest of the term is not bounded by pr --
   //  nothing leaves that way -- and may be beyond a double's range; 0 * inf would be NaN)
   if (wantTrans) {
      const int t0 = __shfl(cTrans, 0);
      const bool uniform = __all(!valid || cTrans == t0);
      if (uniform) {
         const int N0 = __shfl(N, 0);
         double *tr = a.acc + a.lay.tr + a.transOff[t0];
         double *oc = a.acc + a.lay.trOcc + a.trOccOff[t0];
         for (int i = 2; i <= N0 - 1; i++) {
            const bool mine = valid && j == i;
#pragma unroll
            for (int d = 0; d < 5; d++) {
               const int jj = i + d - 2;
               if (jj < 2 || jj > N0 - 1) continue;
               double v = mine ? taOut[d] : 0.0;
#pragma unroll
               for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
               if (lane == 0 && v != 0.0) atomicAdd(tr + (size_t)(i - 1) * N0 + (jj - 1), v);
            }
            double v = mine ? taExit : 0.0, w = mine ? taEntry : 0.0, z = mine ? occJ : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); w += __shfl_xor(w, o); z += __shfl_xor(z, o); }
            if (lane == 0) {
               if (v != 0.0) atomicAdd(tr + (size_t)(i - 1) * N0 + (N0 - 1), v);
               if (w != 0.0) atomicAdd(tr + (size_t)(i - 1), w);
               if (z != 0.0) atomicAdd(oc + (i - 1), z);
            }
         }
         double z = (valid && s.first) ? occE : 0.0;
#pragma unroll
         for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
         if (lane == 0 && z != 0.0) atomicAdd(oc, z);
      } else if (valid) {
         double *tr = a.acc + a.lay.tr + a.transOff[cTrans];
         double *oc = a.acc + a.lay.trOcc + a.trOccOff[cTrans];
#pragma unroll
         for (int d = 0; d < 5; d++) {
            const int jj = j + d - 2;
            if (jj >= 2 && jj <= N - 1 && taOut[d] != 0.0) atomicAdd(tr + (size_t)(j - 1) * N + (jj - 1), taOut[d]);
         }
         if (taExit != 0.0) atomicAdd(tr + (size_t)(j - 1) * N + (N - 1), taExit);
         if (taEntry != 0.0) atomicAdd(tr + (size_t)(j - 1), taEntry);
         if (occJ != 0.0) atomicAdd(oc + (j - 1), occJ);
         if (s.first && occE != 0.0) atomicAdd(oc, occE);
      }
   }
   if (valid && s.first) atomicAdd(a.acc + a.lay.nEgs + cHmm, 1.0);
   if (gl == 0) {
      atomicAdd(a.acc + a.lay.totalPr, pr);
      atomicAdd(a.acc + a.lay.totalT, (double)T);
      atomicAdd(a.acc + a.lay.nUttDone, 1.0);
      atomicAdd(a.acc + a.lay.nEval, (double)ud.nEval);
   }
}

int htkamd_launch_beta_l(const FbArgs &a, int W, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (W == 1) hipLaunchKernelGGL((k_beta_l<1>), dim3(a.nList), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_beta_l<2>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_beta_l<4>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_beta_l<8>), dim3(a.nList), dim3(512), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_launch_alpha_l(const FbArgs &a, int W, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (W == 1) hipLaunchKernelGGL((k_alpha_l<1>), dim3(a.nList), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_alpha_l<2>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_alpha_l<4>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_alpha_l<8>), dim3(a.nList), dim3(512), 0, s, a);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}
