// fb_state.hip -- K2s / K3s: beta and alpha passes with ONE LANE PER CHAIN STATE.
//
// Same reference semantics as fb_wave.hip / fb_kernels.hip (SetBeta HFB.c:1149, StepAlpha :686, InitAlpha :616, MaxModelProb :655,
// SetOcct :399, UpTranParms :1371, UpMixParms seeds :1479 -- S == 1) for utterances WITHOUT tee models whose models have at most
// five states and whose chain has at most 512 emitting states: the everyday case (a 500-frame utterance of 41 three-state models is
// 123 chain states).  Utterances outside that go to the wave-per-model kernels (fb_wave.hip) or the general ones.
//
// Why a second mapping.  With a lane per MODEL (fb_wave.hip) a wavefront steps through ~1 400 instructions per frame -- three
// states' recursions, thirteen transition counters and the beam logic run one after the other in every lane, 41 of 64 lanes live --
// and a batch of 1 250 utterances is 1 250 wavefronts on 1 024 SIMDs: every instruction's issue slot is exposed, the kernels were
// ISSUE-bound at ~9 400 cycles per frame (profiles/r01_pmc_notes.md).  With a lane per emitting STATE the same frame is
// ~150 instructions on 2 wavefronts (123 of 128 lanes live): a state needs its neighbours' values of the previous column (lanes
// l-2..l+2 inside its model), its model's entry value (the exit value of the model before it) and nothing else.  Those travel
// through a few LDS arrays indexed by lane (ds_write own / ds_read lane+d): one write-read round trip per exchange, with a
// workgroup barrier only when the utterance spans more than one wavefront.
//
// Arithmetic is the reference's, operand for operand: the log-adds of a state run over its predecessors (successors) in ascending
// state order, entry term first (alpha) / exit term first (beta); with FAST = false every alpha, beta and the utterance probability
// equal fb_wave.hip's bit for bit (and the oracle's), FAST = true uses the fp32-transcendental LAdd of ladd.h (tolerance class).
//
// Layout of beta for the alpha pass: betaS[(t-1)*L + lane] (emitting states), betaE[(t-1)*L + first lane of model q] = beta_1(q,t);
// L = 64*W.  beta_N(q,t) is not stored: it is a copy of beta_1(q+1,t+1) (or log-zero outside the beam of t+1), re-derived there.
#include <hip/hip_runtime.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "ladd.h"
#include "wavegrp.h"

#include "fb_state.h"

// ------------------------------------------------------------------------------------ K2s: beta
// Tolerance-class kernels (FAST): a term whose transition or predecessor is log-zero needs no test -- ladd_fast adds exp2(-1.4e10) = 0
// to a live value and keeps a dead one below LSMALL, exp_fast of anything below the floor is 0 -- so the lane-divergent branches of the
// exact form (an exec-mask save / branch / restore each: ~200 scalar instructions per step) are not compiled in.
#define LADD_TERM(x, cond, v) do { if constexpr (FAST) x = ladd_fast((x), (v)); else if (cond) x = ladd((x), (v)); } while (0)
#define EXP_TERM(acc, x) do { if constexpr (FAST) acc += exp_fast(x); else if ((x) > EXPFLOOR) acc += EXPT(x); } while (0)

template <int W, bool FAST>
__global__ __launch_bounds__(64 * W) void k_beta_s(FbArgs a)
{
   constexpr int L = 64 * W, LP = L + 2 * SPAD;
   __shared__ double ltab[FAST ? 1 : LADD_TAB_DOUBLES];
   __shared__ double xbeta[2][LP];                     // beta_j of the column just computed, by step parity
   __shared__ double xobs[2][LP];                      // b_j of that column
   __shared__ float stage[W][2 * 64 * 4];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   __shared__ short sqOf[L];                           // model of every lane
   __shared__ short flOf[L + 2];                       // first lane of model q (1-based; [Q+1] = number of chain states)
   if constexpr (!FAST) ladd_table_to_lds(ltab, a.laddTab);
   const int gl = threadIdx.x, lane = gl & 63, wv = gl >> 6;
   // the padding on both sides of the exchange arrays is read by the edge lanes (their transitions there are log-zero): log-zero values,
   // so that the FAST form's untested terms stay dead whatever the LDS held before
   if (gl < 2 * SPAD) { const int p_ = (gl < SPAD) ? gl : L + gl; xbeta[0][p_] = LZERO; xbeta[1][p_] = LZERO; xobs[0][p_] = 0.0; xobs[1][p_] = 0.0; }
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
   for (int i = gl; i < LP; i += L) { xbeta[0][i] = LZERO; xbeta[1][i] = LZERO; xobs[0][i] = 0.0; xobs[1][i] = 0.0; }
   sqOf[gl] = (short)(valid ? s.q : Q + 1);
   if (valid && s.first) flOf[s.q] = (short)gl;
   if (gl == 0) { flOf[Q + 1] = (short)nS; flOf[0] = 0; }
   __syncthreads();
   // which neighbour offsets any state of the utterance uses (wave-uniform: unused ones cost nothing)
   bool useOut[5], useEnt[3];
#pragma unroll
   for (int d = 0; d < 5; d++) useOut[d] = g.ballot(valid && s.aOut[d] > (float)LSMALL).highest() >= 0;
#pragma unroll
   for (int k = 0; k < 3; k++) useEnt[k] = g.ballot(valid && ((s.first && s.aEntryOf[k] > (float)LSMALL) || aEntryNext[k] > (float)LSMALL)).highest() >= 0;
   const int q = s.q, N = s.N, j = s.j;
   const int offNext = N - j;                          // lane of the next model's first state = gl + offNext
   const short *tLo = a.taperLo + ud.frame0 - 1, *tHi = a.taperHi + ud.frame0 - 1;   // 1-based t
   short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;
   ObsRow st;
   st.lds = stage[wv]; st.lane = lane; st.R = (f4s)(0.f);
   st.row = valid ? a.outp + ud.outp0 + (size_t)gl * T : nullptr;
   const double mle = a.minLogExp;
   const bool pruning = a.pruneInit < HTKAMD_NOPRUNE;
#define ladd(x, y) ladd_sel<FAST>((x), (y), mle, ltab)
   // everything a step needs of the column before it, read in ONE pass after the single exchange of a step:
   //   ySucc/oSucc[d]  beta and score of the own model's states j+d-2        bE   beta_1 of the own model (first lane; also lMax)
   //   bEn             beta_1 of the NEXT model = this model's exit value of the step to come
#define BETA_GATHER(par, wantOwn)                                                                                                   \
   do {                                                                                                                           \
      const double *xb_ = xbeta[par] + SPAD + gl, *xo_ = xobs[par] + SPAD + gl;                                                    \
      _Pragma("unroll") for (int d = 0; d < 5; d++) if (useOut[d]) { ySucc[d] = xb_[d - 2]; oSucc[d] = xo_[d - 2]; }              \
      double x_ = LZERO;                                                                                                          \
      _Pragma("unroll") for (int k = 0; k < 3; k++)                                                                               \
         if (useEnt[k]) {                                                                                                         \
            const double aa = aEntryNext[k], y = xb_[offNext + k];                                                                \
            LADD_TERM(x_, aa > LSMALL && y > LSMALL, aa + xo_[offNext + k] + y);                                                   \
         }                                                                                                                        \
      bEn = x_;                                                                                                                   \
      lMax = LZERO;                                                                                                               \
      if ((wantOwn) && s.first) {                                                                                                 \
         x_ = LZERO;                                                                                                              \
         _Pragma("unroll") for (int k = 0; k < 3; k++)                                                                            \
            if (2 + k <= N - 1) {                                                                                                 \
               const double aa = s.aEntryOf[k], y = xb_[k];                                                                       \
               if (y > lMax) lMax = y;                                                                                            \
               LADD_TERM(x_, aa > LSMALL && y > LSMALL, aa + xo_[k] + y);                                                          \
            }                                                                                                                     \
         bE = x_;                                                                                                                 \
      }                                                                                                                           \
   } while (0)

   double thresh = a.pruneInit, pr = LZERO;
   int ok = 0;
   for (;;) {                                            // StepBack retry loop (HFB.c:1332-1361)
      int fail = 0;
      double bJ = LZERO, bE = LZERO, bEn = LZERO, lMax = LZERO;   // beta_j(t); beta_1(q,t) (first lane); beta_1(q+1,t); max_j beta_j(t)
      double ySucc[5]; float obT = 0.f, obP = 0.f;
      double oSucc[5];
#pragma unroll
      for (int d = 0; d < 5; d++) { ySucc[d] = LZERO; oSucc[d] = 0.0; }
      // ---- t = T (HFB.c:1175-1198): only the last model can end the utterance (no tee models in this path)
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
      const bool inT = valid && q >= endT;
      {
         // exit value of model q at T: 0 for the last model, log-zero sums for the others (e(q) = e(q+1) + a_1N(q+1), a_1N = LZERO)
         double mine = 0.0;
         for (int k = Q; k > q && k > endT; k--) mine += (double)(float)LZERO;
         if (inT) bJ = (double)s.aExit + mine;
         xbeta[T & 1][SPAD + gl] = inT ? bJ : LZERO; xobs[T & 1][SPAD + gl] = (double)obT;
         xsync<W>();
         BETA_GATHER(T & 1, inT);
         if (inT) { BETA_S(T) = bJ; if (s.first) BETA_E(T) = bE; }
      }
      if (gl == 0) { gLo[T] = (short)endT; gHi[T] = (short)Q; }
      int qHiN = Q, qLoN = endT, lastEnd = endT;
      int nxtLo = (T >= 2) ? tLo[T - 1] : 1, nxtHi = (T >= 2) ? tHi[T - 1] : 1;
      bool stPrev = false, stIn = false; int tPrev = 0, loPrev = 1, hiPrev = 1;

      // ---- t = T-1 .. 1 (HFB.c:1205-1277)
      for (int t = T - 1; t >= 1; t--) {
         const int taperLoT = nxtLo, taperHiT = nxtHi;
         if (t >= 2) { nxtLo = tLo[t - 1]; nxtHi = tHi[t - 1]; }
         if (t >= 2 && ((t - 2) & 3) == 3) { st.park((t - 2) >> 2); if (((t - 2) >> 2) >= 1) st.load(((t - 2) >> 2) - 1); }
         if (stPrev) {                                   // the column finished in the previous iteration goes out now
            if (stIn) { BETA_S(tPrev) = bJ; if (s.first) BETA_E(tPrev) = bE; }
            if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; }
            stPrev = false;
         }
         obT = obP;
         if (t >= 2) obP = st.get(t - 2);
         const int startq = qHiN;
         const int endq = (qLoN == 1) ? 1 : ((taperLoT >= qLoN) ? taperLoT : qLoN - 1);
         const bool inRange = valid && q >= endq && q <= startq;
         const bool wasIn = q >= qLoN && q <= qHiN;
         if (inRange) {
            const bool p1 = (q < Q) && (q + 1 >= qLoN) && (q + 1 <= qHiN);
            const double ex = p1 ? bEn : LZERO;                  // beta_N(q,t) = beta_1(q+1,t+1)
            double x = (double)s.aExit + ex;
            if (wasIn) {
#pragma unroll
               for (int d = 0; d < 5; d++)
                  if (useOut[d]) {
                     const double aa = s.aOut[d], y = ySucc[d];
                     LADD_TERM(x, aa > LSMALL && y > LSMALL, aa + oSucc[d] + y);
                  }
            }
            bJ = x;
         }
         // the one exchange of the step: publish the new column (buffers alternate with the step's parity, so the writes of this
         // step never meet the reads of the step before), gather everything the next step needs of it
         xbeta[t & 1][SPAD + gl] = inRange ? bJ : LZERO; xobs[t & 1][SPAD + gl] = (double)obT;
         xsync<W>();
         BETA_GATHER(t & 1, inRange);
         int newHi, newLo;
         if (!pruning) {                                 // only the taper acts (HFB.c:1259-1264)
            newHi = (taperHiT < startq) ? taperHiT : startq;
            newLo = endq;
         } else {                                        // beam pruning (HFB.c:1254-1272): one bit per model, at its first lane
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
         if (stIn) { BETA_S(tPrev) = bJ; if (s.first) BETA_E(tPrev) = bE; }
         if (gl == 0) { gLo[tPrev] = (short)loPrev; gHi[tPrev] = (short)hiPrev; }
      }
      if (!fail) {
         pr = g.bcast(bE, flOf[lastEnd]);                // utt->pr = beta_1 of the last model processed
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
#undef BETA_GATHER
#undef ladd
}

// ------------------------------------------------------------------------------------ K3s: alpha + stats
template <int W, bool FAST>
__global__ __launch_bounds__(64 * W) void k_alpha_s(FbArgs a)
{
   constexpr int L = 64 * W, LP = L + 2 * SPAD;
   __shared__ double ltab[FAST ? 1 : LADD_TAB_DOUBLES];
   __shared__ double etab[FAST ? 1 : EXP_TAB_N];
   __shared__ double xalpha[2][LP];                    // alpha_j(t) by step parity
   __shared__ double xsum[2][LP];                      // alpha_j(t) + beta_j(t) inside the beta beam (MaxModelProb)
   __shared__ double xnext[2][LP];                     // b_j(t+1) + beta_j(t+1) inside the beam of t+1 (transition counts)
   __shared__ float stage[W][2 * 64 * 4];
   __shared__ unsigned long long gx[2 * W * 4];
   __shared__ float ga1[64 * W];
   __shared__ short sqOf[L];
   __shared__ short flOf[L + 2];
   if constexpr (!FAST) { ladd_table_to_lds(ltab, a.laddTab); exp_table_to_lds(etab); }
   const int gl = threadIdx.x, lane = gl & 63, wv = gl >> 6;
   if (gl < 2 * SPAD) {                                // padding of the exchange arrays: log-zero (see k_beta_s)
      const int p_ = (gl < SPAD) ? gl : L + gl;
      xalpha[0][p_] = LZERO; xalpha[1][p_] = LZERO; xsum[0][p_] = LZERO; xsum[1][p_] = LZERO; xnext[0][p_] = LZERO; xnext[1][p_] = LZERO;
   }
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
      for (int k = 0; k < 2; k++) { xalpha[k][i] = LZERO; xsum[k][i] = LZERO; xnext[k][i] = LZERO; }
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
   const MaskW<W> firsts = g.ballot(valid && s.first);   // loop-invariant: one bit per model, at its first lane
   const int q = s.q, N = s.N, j = s.j;
   const int offNext = N - j, offPrev = -(j - 1);       // first lane of the next model / last lane of the previous model
   const int cHmm = valid ? a.mHmm[s.mi] : 0, cTrans = valid ? a.mTrans[s.mi] : 0, mc0 = valid ? a.mCell0[s.mi] : 0;
   int cM = 0;
   if (valid) { const int sidx = a.slotState[ud.slot0 + gl]; cM = a.stateCompOff[sidx + 1] - a.stateCompOff[sidx]; }
   const short *gLo = a.qLo + ud.frame0 - 1, *gHi = a.qHi + ud.frame0 - 1;    // final beta beam, 1-based t
   short *gaLo = a.aLo + ud.frame0 - 1, *gaHi = a.aHi + ud.frame0 - 1;
   ObsRow st;
   st.lds = stage[wv]; st.lane = lane; st.R = (f4s)(0.f);
   st.row = valid ? a.outp + ud.outp0 + (size_t)gl * T : nullptr;
   double *gam = a.gam + ud.gam0 + gl;
   // beta_1 of the NEXT model (its first lane's column of betaE): every lane of a model reads the same word
   const double *bNextE = a.betaW + ud.betaW0 + (size_t)T * L + (gl + offNext);
   const bool hasNext = valid && q < Q;
   const double mle = a.minLogExp, pr = a.pr[u];
   const double minF = (double)a.minFrwdP;
   const bool wantMix = (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES)) != 0;
   const bool wantTrans = (a.uFlags & HTKAMD_UPTRANS) != 0;
#define ladd(x, y) ladd_sel<FAST>((x), (y), mle, ltab)
#define EXPT(x) exp_sel<FAST>((x), etab)

   double aJ = LZERO, aE = LZERO, aEnext = LZERO;        // alpha_j(t); alpha_1(q,t); alpha_1(q,t+1) = exit value of the model before, column t
   double yIn[5];                                        // alpha of the own model's states j+d-2 in the column before
   double xpre = LZERO;
   double taOut[5], taExit = 0.0, taEntry = 0.0, occJ = 0.0, occE = 0.0;
#pragma unroll
   for (int d = 0; d < 5; d++) { taOut[d] = 0.0; yIn[d] = LZERO; }
   // beta / scores / beams of frames t, t+1 and (in flight) t+2
   double bT = LZERO, bT1 = LZERO, bT2 = LZERO, eT = LZERO, eT1 = LZERO, eT2 = LZERO, nT1 = LZERO, nT2 = LZERO;   // nT1 = beta_1(q+1,t+1)
   float oT = 0.f, oT1 = 0.f;
   int lo0 = 1, hi0 = 0, lo1 = gLo[1], hi1 = gHi[1], lo2 = (T >= 2) ? gLo[2] : 1, hi2 = (T >= 2) ? gHi[2] : 0, lo3 = 1, hi3 = 0;
   if (valid) {
      bT = BETA_S(1); if (s.first) eT = BETA_E(1);
      if (T >= 2) { bT1 = BETA_S(2); if (s.first) eT1 = BETA_E(2); if (hasNext) nT1 = bNextE[(size_t)1 * L]; }
   }
   st.load(0); st.park(0);
   if (T > 4) st.load(1);
   oT = st.get(0); if (T >= 2) oT1 = st.get(1);
   int sq = 1, eq = hi1, err = 0;
   double mmpA = LZERO;                                  // MaxModelProb of this model in the column just finished (first lane)

   for (int t = 1; t <= T; t++) {
      // request column t+2
      if (t + 2 <= T) {
         lo3 = gLo[t + 2]; hi3 = gHi[t + 2];
         if (valid) { bT2 = BETA_S(t + 2); if (s.first) eT2 = BETA_E(t + 2); if (hasNext) nT2 = bNextE[(size_t)(t + 1) * L]; }
      }
      bool in;
      if (t == 1) {
         // ---- InitAlpha (HFB.c:616-651): without tee models only the first model starts
         double a1 = 0.0;
         for (int k = 2; k <= q && k <= eq; k++) a1 += (double)(float)LZERO;
         in = valid && q <= eq;
         if (in) {
            aE = a1;
            const double aa = s.aEntry;
            xpre = aE + aa;
            aJ = (aa > LSMALL) ? xpre + (double)oT : LZERO;
         }
      } else {
         // ---- alpha beam (HFB.c:699-722) from MaxModelProb of column t-1: one bit per model at its first lane
         const MaskW<W> kept = firsts & ~g.ballot(valid && s.first && (pr - mmpA > minF));
         // first model >= qLo[t-1] that is kept; running past the chain "keeps" (mirrors the reference's scan)
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
         // ---- alpha column t (HFB.c:729-771) from the values gathered after the previous step's exchange
         in = valid && q >= sq && q <= eq;
         if (valid && !in) { aJ = LZERO; aE = LZERO; }
         if (in) {
            const double a1 = (q == 1) ? LZERO : aEnext;             // alpha_1(q,t) = alpha_N(q-1,t-1)
            aE = a1;
            double aa = s.aEntry;
            double x = (aa > LSMALL) ? aa + a1 : LZERO;
#pragma unroll
            for (int d = 0; d < 5; d++)
               if (useIn[d]) {
                  aa = s.aIn[d];
                  const double y = yIn[d];
                  LADD_TERM(x, aa > LSMALL && y > LSMALL, y + aa);
               }
            xpre = x;
            aJ = x + (double)oT;
         }
      }
      if (gl == 0) { gaLo[t] = (short)sq; gaHi[t] = (short)eq; }

      // ---- the one exchange of the step: alpha_j(t), alpha_j + beta_j (MaxModelProb), b_j + beta_j of t+1 (transition counts)
      const bool inB = valid && q >= lo1 && q <= hi1;    // in the beta beam of t
      const bool inBeam = valid && q >= sq && q <= eq;
      const bool bqt1ok = (t < T) && q >= lo2 && q <= hi2;
      const int par = t & 1;
      xalpha[par][SPAD + gl] = in ? aJ : LZERO;
      xsum[par][SPAD + gl] = inB ? aJ + bT : LZERO;
      xnext[par][SPAD + gl] = bqt1ok ? (double)oT1 + bT1 : LZERO;
      xsync<W>();
      const double *xa = xalpha[par] + SPAD + gl;
      // own model's states for the next column
#pragma unroll
      for (int d = 0; d < 5; d++) if (useIn[d]) yIn[d] = xa[d - 2];
      // exit value of the model BEFORE this one in column t (HFB.c:762-769 there): alpha_1 of this model in column t+1
      double aXp = LZERO;
      if (valid && q > 1) {
#pragma unroll
         for (int k = 2; k >= 0; k--)
            if (useExitP[k]) {
               const double aa = aExitPrev[k], y = xa[offPrev - k];
               LADD_TERM(aXp, aa > LSMALL && y > LSMALL, y + aa);
            }
      }
      if (a.alphaDbg && valid) {
         double *ad = a.alphaDbg + ud.beta0 + (size_t)(t - 1) * nC + mc0;
         ad[j - 1] = aJ;
         if (s.first) ad[0] = aE;
         if (s.last) {                                   // debugging aid only: this model's own exit value
            double x = LZERO;
#pragma unroll
            for (int k = 2; k >= 0; k--)
               if (N - 1 - k >= 2) {
                  const double aa = s.aExitOf[k], y = xa[-k];
                  if (aa > LSMALL && (t == 1 || y > LSMALL)) x = ladd(x, y + aa);
               }
            ad[N - 1] = x;
         }
      }
      // ---- statistics for column t (HFB.c:1790-1806) and MaxModelProb of column t
      // beta_N(q,t): 0 for the last model at T, else beta_1(q+1,t+1) where that is in the beam of t+1
      double bN = LZERO;
      if (valid) bN = (t == T) ? ((q == Q) ? 0.0 : LZERO) : ((hasNext && q + 1 >= lo2 && q + 1 <= hi2) ? nT1 : LZERO);
      if (valid && s.first) {
         double mm = LZERO;
         if (inB) {
            mm = aE + eT;                                // i = 1
            const double *xs = xsum[par] + SPAD + gl;
#pragma unroll
            for (int k = 0; k < 3; k++) if (2 + k <= N - 1) { const double v = xs[k]; if (v > mm) mm = v; }
         }
         // alpha_N + beta_N of the model before (HFB.c:662-666); its beta_N(t) is this model's beta_1(t+1) inside the beam of t+1
         double prevExit = LZERO;
         if (q > 1 && q - 1 >= lo1 && q - 1 <= hi1) {
            const double bNp = (t == T) ? LZERO : ((q >= lo2 && q <= hi2) ? eT1 : LZERO);
            prevExit = aXp + bNp;
         }
         mmpA = (prevExit > mm) ? prevExit : mm;
      }
      if (inBeam) {
         // SetOcct (HFB.c:399-418) + UpTranParms (HFB.c:1390-1410) for state j, and for the entry state in the model's first lane
         double x = aJ + bT - pr;
         occJ += (double)((x > EXPFLOOR) ? (float)EXPT(x) : 0.0f);
         if (s.first) {
            x = aE + eT - pr;
            occE += (double)((x > EXPFLOOR) ? (float)EXPT(x) : 0.0f);
         }
         if (wantTrans) {
            x = aE + (double)s.aEntry + (double)oT + bT - pr;
            EXP_TERM(taEntry, x);
            if (bqt1ok) {
               const double *xn = xnext[par] + SPAD + gl;
#pragma unroll
               for (int d = 0; d < 5; d++)
                  if (useOut[d]) {
                     x = aJ + (double)s.aOut[d] + xn[d - 2] - pr;
                     EXP_TERM(taOut[d], x);
                  }
            }
            x = aJ + (double)s.aExit + bN - pr;
            EXP_TERM(taExit, x);
         }
      }
      if (valid) {
         // UpMixParms seed (HFB.c:1479-1489,1573-1606)
         double seed = LZERO;
         if (inBeam && wantMix) {
            if (cM == 1 || a.maxM == 1) {
               const double x = aJ + bT - pr;
               if (-x < minF) seed = x;
            } else {
               const double initx = xpre + (bT - pr);
               const double ub = initx + (double)oT;
               if (ub > -minF - 0.01) seed = initx;
            }
         }
         gam[(size_t)(t - 1) * nS] = seed;
      }
      aEnext = aXp;
      // rotate: t -> t+1
      bT = bT1; bT1 = bT2; eT = eT1; eT1 = eT2; nT1 = nT2;
      oT = oT1;
      if (t + 2 <= T) {
         const int f = t + 1;                            // frame index (0-based) of t+2
         if ((f & 3) == 0) { st.park(f >> 2); if (4 * ((f >> 2) + 1) < T) st.load((f >> 2) + 1); }
         oT1 = st.get(f);
      }
      lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2; lo2 = lo3; hi2 = hi3;
   }

   if (err) {
      if (gl == 0) { a.status[u] = HTKAMD_UTT_EALPHA; atomicAdd(a.acc + a.lay.nUttSkipped, 1.0); }
      return;
   }
   // ---- flush.  With one transition matrix for the whole chain (a tied-transition system) the counts of equal (i,j) are summed
   // over the lanes first: one atomic per matrix entry and wavefront.
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
#undef ladd
#undef EXPT
}

template <bool FAST> static void launch_beta_s(const FbArgs &a, int W, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_beta_s<1, FAST>), dim3(a.nList), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_beta_s<2, FAST>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_beta_s<4, FAST>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_beta_s<8, FAST>), dim3(a.nList), dim3(512), 0, s, a);
}
template <bool FAST> static void launch_alpha_s(const FbArgs &a, int W, hipStream_t s)
{
   if (W == 1) hipLaunchKernelGGL((k_alpha_s<1, FAST>), dim3(a.nList), dim3(64), 0, s, a);
   else if (W == 2) hipLaunchKernelGGL((k_alpha_s<2, FAST>), dim3(a.nList), dim3(128), 0, s, a);
   else if (W == 4) hipLaunchKernelGGL((k_alpha_s<4, FAST>), dim3(a.nList), dim3(256), 0, s, a);
   else hipLaunchKernelGGL((k_alpha_s<8, FAST>), dim3(a.nList), dim3(512), 0, s, a);
}

// a.uttList / a.nList: the utterances of one class (W wavefronts of 64 chain states each)
int htkamd_launch_beta_s(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (fast) launch_beta_s<true>(a, W, s); else launch_beta_s<false>(a, W, s);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

int htkamd_launch_alpha_s(const FbArgs &a, int W, bool fast, hipStream_t s)
{
   if (a.nList <= 0) return HTKAMD_OK;
   if (fast) launch_alpha_s<true>(a, W, s); else launch_alpha_s<false>(a, W, s);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}
