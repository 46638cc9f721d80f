// viterbi.hip -- K5: Viterbi forced alignment (HVite -a) of a batch of utterances, one wavefront per utterance.
//
// Reference semantics (HRec.c, 1-best, alignment network = linear chain of the transcription's models, see
// oracle/orc_viterbi.c for the restatement this kernel is tested against):
//   StepHMM1 642-787   best predecessor over [seIndex[j][0], seIndex[j][1]], first maximum wins; token likes are
//                      double sums of float log transition / output probabilities; genThresh state pruning
//   ProcessObservation 1935-2030  genThresh = float(genMax - genBeam) floored at LSMALL, one frame stale in pass 1;
//                      pass 2 detaches instances below it, StepHMM2 (790) passes tee models, SetEntryState (1303)
//   LatFromPaths 1512-1660 / TranscriptionFromLattice 2176  state and model segments with scores
//
// MI355X mapping: as in fb_wave.hip, lane q owns model q (tokens, transition matrix, seIndex in registers); the
// exit -> entry hand-over to the next model is a wave shuffle, the per-frame maximum a wave reduction, the tee
// chain a loop over the set bits of a ballot.  Output probabilities come from K1 (bit-exact scores), so the token
// likes -- the same additions in the same order as the reference -- are bit-identical and so are the alignments.
// Per frame each lane writes its states' back-pointers (1 byte) and pre-output likes (8 bytes) as a contiguous run.
// A second tiny kernel walks the back-pointers (one lane per utterance) and emits the segments.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "wavegrp.h"

#define UPB(W) ((W) == 1 ? 4 : 1)   // utterances per workgroup: four single-wave utterances, or one multi-wave utterance
#define VMAXN 5

struct VitUtt {
   int T, Q, nSlots, frame0, q0, slot0, status, pad;
   size_t outp0;      // floats : outp[outp0 + slot*T + (t-1)]
   size_t tr0;        // per-(t,slot) trellis index base: (t-1)*nSlots + slot
   size_t mt0;        // per-(t,model) trellis index base: t*Q + (q-1), t = 0..T
   size_t seg0;       // per-slot output base
   size_t mod0;       // per-model output base
};

struct VitArgs {
   const VitUtt *utt; int nUtt;
   const int *uttList; int nList;          // the utterances of this launch (one class: same number of wavefronts per utterance)
   const int *mN, *mTp, *mSlot0;
   const float *transP, *outp;
   signed char *bp; double *pre;           // [sum T*nSlots]
   double *exl, *entAt; signed char *exbp; // [sum (T+1)*Q]
   int *segStart, *segEnd; double *segScore;   // [sum nSlots]
   int *modStart, *modEnd; double *modScore;   // [sum Q]
   double *total; int *status;             // [nUtt]
   float genBeam;
};

template <int MAXN, int W>
__global__ __launch_bounds__(64 * W * UPB(W)) void k_viterbi_w(VitArgs a)
{
   __shared__ unsigned long long gx[2 * W * 8];
   const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
   const int li = blockIdx.x * UPB(W) + wib / W;         // all wavefronts of an utterance leave together
   if (li >= a.nList) return;
   const int u = a.uttList[li];
   Grp<W> g; g.x = gx; g.a1 = nullptr; g.wave = wib % W; g.lane = lane; g.ph = 0;
   const int gl = 64 * g.wave + lane;                    // group lane: model gl+1 of the chain
   const VitUtt ud = a.utt[u];
   if (ud.status != HTKAMD_UTT_OK) { if (gl == 0) { a.status[u] = ud.status; a.total[u] = LZERO; } return; }
   const int T = ud.T, Q = ud.Q, nSlots = ud.nSlots;
   const int q = gl + 1;
   const bool valid = q <= Q;
   int N = 0, ms0 = 0;
   float tp[MAXN][MAXN];
   int seLo[MAXN + 1], seHi[MAXN + 1];
#pragma unroll
   for (int i = 0; i < MAXN; i++)
#pragma unroll
      for (int j = 0; j < MAXN; j++) tp[i][j] = (float)LZERO;
#pragma unroll
   for (int j = 0; j <= MAXN; j++) { seLo[j] = 1; seHi[j] = 1; }
   if (valid) {
      const int mi = ud.q0 + q - 1;
      N = a.mN[mi]; ms0 = a.mSlot0[mi];
      const float *gt = a.transP + a.mTp[mi];
#pragma unroll
      for (int i = 0; i < MAXN; i++)
#pragma unroll
         for (int j = 0; j < MAXN; j++)
            if (i < N && j < N) tp[i][j] = gt[i * N + j];
      // CreateSEIndex (HRec.c:1403-1431)
#pragma unroll
      for (int j = 2; j <= MAXN; j++)
         if (j <= N) {
            int mn = N, mx = 1;
#pragma unroll
            for (int i = MAXN - 1; i >= 1; i--)
               if (i < N && i >= ((j == N) ? 2 : 1) && tp[i - 1][j - 1] > (float)LSMALL) mn = i;
#pragma unroll
            for (int i = 2; i < MAXN; i++)
               if (i < N && tp[i - 1][j - 1] > (float)LSMALL) mx = i;
            if (mn > mx) { mn = (j == N) ? 2 : 1; mx = N - 1; }
            seLo[j] = mn; seHi[j] = mx;
         }
   }
   float aN[MAXN];                                      // aN[i] = log a_iN
   int loN = 2, hiN = 1;
#pragma unroll
   for (int i = 0; i < MAXN; i++) aN[i] = (float)LZERO;
#pragma unroll
   for (int j = 2; j <= MAXN; j++)
      if (j == N) {
         loN = seLo[j]; hiN = seHi[j];
#pragma unroll
         for (int i = 1; i < MAXN; i++) if (i < N) aN[i] = tp[i - 1][j - 1];
      }
   const float a1N = aN[1 % MAXN];
   const bool tee = valid && a1N > (float)LSMALL;
   const MaskW<W> teeMask = g.ballot(tee);

   const float *orow = a.outp + ud.outp0 + (size_t)ms0 * T;
   signed char *gbp = a.bp + ud.tr0 + ms0;
   double *gpre = a.pre + ud.tr0 + ms0;
   double *gexl = a.exl + ud.mt0 + (q - 1), *gent = a.entAt + ud.mt0 + (q - 1);
   signed char *gexbp = a.exbp + ud.mt0 + (q - 1);

   double like[MAXN + 1];                                // like[i], i = 1 (entry) .. N-1
#pragma unroll
   for (int i = 0; i <= MAXN; i++) like[i] = LZERO;
   double exitL = LZERO, instMax = LZERO;
   bool active = false;
   float genThresh = (float)LSMALL;
   float ob[MAXN], obN[MAXN];
#pragma unroll
   for (int j = 0; j < MAXN; j++) { ob[j] = 0.f; obN[j] = 0.f; }
   if (valid && T >= 1) {
#pragma unroll
      for (int j = 2; j < MAXN; j++) if (j < N) obN[j] = orow[(size_t)(j - 2) * T];
   }
   double finalLike = LZERO;

   for (int t = 0; t <= T; t++) {
      double genMax = LZERO;
      int exArg = 0;
      if (t >= 1) {
#pragma unroll
         for (int j = 0; j < MAXN; j++) ob[j] = obN[j];
         if (valid && t + 1 <= T) {
#pragma unroll
            for (int j = 2; j < MAXN; j++) if (j < N) obN[j] = orow[(size_t)(j - 2) * T + t];
         }
         // ---- pass 1: StepHMM1
         if (valid && active) {
            double nw[MAXN + 1];
            double mx = LZERO;
#pragma unroll
            for (int j = 2; j < MAXN; j++) {
               nw[j] = LZERO;
               if (j < N) {
                  int arg = seLo[j];
                  double best = LZERO;
#pragma unroll
                  for (int i = 1; i < MAXN; i++) if (i == arg) best = like[i] + (double)tp[i - 1][j - 1];
#pragma unroll
                  for (int i = 2; i < MAXN; i++)
                     if (i > seLo[j] && i <= seHi[j]) {
                        const double c = like[i] + (double)tp[i - 1][j - 1];
                        if (c > best) { best = c; arg = i; }
                     }
                  gpre[(size_t)(t - 1) * nSlots + j - 2] = best;
                  gbp[(size_t)(t - 1) * nSlots + j - 2] = (signed char)arg;
                  if (best > genThresh) {
                     nw[j] = best + (double)ob[j];
                     if (nw[j] > mx) mx = nw[j];
                  }
               }
            }
            like[1] = LZERO;
#pragma unroll
            for (int j = 2; j < MAXN; j++) if (j < N) like[j] = nw[j];
            instMax = mx;
            genMax = mx;
            {  // exit state: best of like_i + a_iN over seIndex[N], first maximum wins (HRec.c:738-760)
               double best = LZERO;
               exArg = 0;
#pragma unroll
               for (int i = 1; i < MAXN; i++)
                  if (i < N && i >= loN && i <= hiN) {
                     const double c = like[i] + (double)aN[i];
                     if (exArg == 0 || c > best) { best = c; exArg = i; }
                  }
               if (best > LSMALL) exitL = best; else { exitL = LZERO; exArg = 0; }
            }
         } else { exitL = LZERO; }
         genMax = g.maxall(genMax);
         genThresh = (float)(genMax - (double)a.genBeam);
         if (genThresh < (float)LSMALL) genThresh = (float)LSMALL;
      }
      // ---- pass 2: exit -> entry of the next model, in chain order
      // A: candidates from the predecessor's pass-1 exit (valid wherever the predecessor is not a tee model)
      const bool selfAlive = valid && active && !(instMax < (double)genThresh);     // not detached on its own account
      double exOut = (selfAlive && exitL > (double)genThresh) ? exitL : LZERO;
      double cand = g.up1(exOut);
      if (q == 1) cand = (t == 0) ? 0.0 : LZERO;
      bool gotEntry = false;
      if (valid && cand > (double)genThresh) {
         if (!active) { active = true; instMax = LZERO; exitL = LZERO;
#pragma unroll
            for (int i = 0; i <= MAXN; i++) like[i] = LZERO; }
         if (cand > like[1]) like[1] = cand;
         if (like[1] > instMax) instMax = like[1];
         gotEntry = true;
      }
      // B: tee models in ascending order: entry -> exit within the frame (StepHMM2), then onward
      // With several wavefronts each one works through the tee models of its own 64 lanes with wave shuffles; only a tee model in
      // the first or last lane of a wavefront involves the neighbouring wavefront (one LDS exchange, all wavefronts take part), and
      // the ascending order across wavefronts is kept by exactly those exchanges.
#pragma unroll
      for (int k = 0; k < W; k++) {
      unsigned long long tm = teeMask.w[k];
      while (tm) {
         const int tlw = __ffsll((long long)tm) - 1, tl = 64 * k + tlw;   // lane / group lane of the tee model
         tm &= tm - 1;
         const bool mine = (W == 1) || g.wave == k;
         // the tee lane refreshes its entry from its predecessor's CURRENT exit (the predecessor may be a tee lane done earlier)
         const bool pAlive = valid && active && !(instMax < (double)genThresh);
         const double pOut = (pAlive && exitL > (double)genThresh) ? exitL : LZERO;
         double c2;
         if (tl == 0) c2 = (t == 0) ? 0.0 : LZERO;
         else if (W > 1 && tlw == 0) c2 = g.bcast(pOut, tl - 1);
         else c2 = __shfl(pOut, (tlw - 1) & 63);
         if (mine && lane == tlw) {
            if (c2 > (double)genThresh) {
               if (!active) { active = true; instMax = LZERO; exitL = LZERO;
#pragma unroll
                  for (int i = 0; i <= MAXN; i++) like[i] = LZERO; }
               if (c2 > like[1]) like[1] = c2;
               if (like[1] > instMax) instMax = like[1];
               gotEntry = true;
            }
            if (active && !(instMax < (double)genThresh)) {
               const double c = like[1] + (double)a1N;
               if (c > exitL) { exitL = c; exArg = 1; }
            }
         }
         // the model after the tee model sees the updated exit
         const bool tAlive = valid && active && !(instMax < (double)genThresh);
         const double tOut = (tAlive && exitL > (double)genThresh) ? exitL : LZERO;
         const bool crossOut = W > 1 && tlw == 63 && k + 1 < W;           // the next model lives in the next wavefront
         const double c3 = crossOut ? g.bcast(tOut, tl) : __shfl(tOut, tlw);
         const bool consumer = crossOut ? (g.wave == k + 1 && lane == 0) : (mine && lane == tlw + 1);
         if (consumer && valid && !teeMask.bit(gl) && c3 > (double)genThresh) {
            if (!active) { active = true; instMax = LZERO; exitL = LZERO;
#pragma unroll
               for (int i = 0; i <= MAXN; i++) like[i] = LZERO; }
            if (c3 > like[1]) like[1] = c3;
            if (like[1] > instMax) instMax = like[1];
            gotEntry = true;
         }
      }
      }
      // detach (HRec.c:2008-2011): nothing alive and nothing arrived
      if (valid && active && instMax < (double)genThresh) {
         active = false; exitL = LZERO; exArg = 0;
#pragma unroll
         for (int i = 0; i <= MAXN; i++) like[i] = LZERO;
      }
      if (valid) {
         gent[(size_t)t * Q] = active ? like[1] : LZERO;
         gexl[(size_t)t * Q] = active ? exitL : LZERO;
         gexbp[(size_t)t * Q] = (signed char)exArg;
      }
      if (t == T) {
         const bool lastAlive = valid && active;
         const double fo = (lastAlive && exitL > (double)genThresh && exitL > LSMALL) ? exitL : LZERO;
         finalLike = g.bcast(fo, Q - 1);
      }
      (void)gotEntry;
   }
   if (gl == 0) {
      a.total[u] = finalLike;
      a.status[u] = (finalLike > LSMALL) ? HTKAMD_UTT_OK : HTKAMD_UTT_SKIPPED;
   }
}

// ------------------------------------------------------------------------------------ general form
// Chains of more than 64 models or models of more than VMAXN states: one workgroup per utterance, thread = model for
// pass 1 (tokens in LDS), thread 0 walks the exit -> entry chain of pass 2 (it is sequential in the reference too:
// instance-list order).  Same arithmetic, same trellis layout as k_viterbi_w, so k_viterbi_trace serves both.
#define VG_THREADS 256
#define VG_MAXN 16
__global__ __launch_bounds__(VG_THREADS) void k_viterbi_g(VitArgs a)
{
   extern __shared__ double vsm[];
   __shared__ double red[VG_THREADS / 64];
   __shared__ float gthr;
   const int u = a.uttList[blockIdx.x], tid = threadIdx.x;
   if (u >= a.nUtt) return;
   const VitUtt ud = a.utt[u];
   if (ud.status != HTKAMD_UTT_OK) { if (tid == 0) { a.status[u] = ud.status; a.total[u] = LZERO; } return; }
   const int T = ud.T, Q = ud.Q, nSlots = ud.nSlots;
   int maxN = 0;
   for (int q = 0; q < Q; q++) { const int n = a.mN[ud.q0 + q]; if (n > maxN) maxN = n; }
   const int LS = maxN + 1;
   double *like = vsm;                                   // like[(q-1)*LS + i], i = 1..N-1
   double *exitL = like + (size_t)Q * LS, *instMax = exitL + Q;
   signed char *exArgS = (signed char *)(instMax + Q), *active = exArgS + Q;
   for (int k = tid; k < Q * LS; k += VG_THREADS) like[k] = LZERO;
   for (int k = tid; k < Q; k += VG_THREADS) { exitL[k] = LZERO; instMax[k] = LZERO; exArgS[k] = 0; active[k] = 0; }
   if (tid == 0) gthr = (float)LSMALL;
   __syncthreads();
   signed char *gbp = a.bp + ud.tr0; double *gpre = a.pre + ud.tr0;
   double *gexl = a.exl + ud.mt0, *gent = a.entAt + ud.mt0; signed char *gexbp = a.exbp + ud.mt0;
   double finalLike = LZERO;

   for (int t = 0; t <= T; t++) {
      if (t >= 1) {
         const float genThresh = gthr;                    // previous frame's threshold
         double myMax = LZERO;
         for (int q = tid; q < Q; q += VG_THREADS) {
            if (!active[q]) { exitL[q] = LZERO; exArgS[q] = 0; continue; }
            const int N = a.mN[ud.q0 + q], ms0 = a.mSlot0[ud.q0 + q];
            const float *tp = a.transP + a.mTp[ud.q0 + q];
            double *lk = like + (size_t)q * LS;
            double nw[VG_MAXN];
            double mx = LZERO;
            for (int j = 2; j < N; j++) {
               int lo = 1, hi = N - 1;                    // CreateSEIndex (HRec.c:1403)
               while (lo < N && !(tp[(lo - 1) * N + (j - 1)] > (float)LSMALL)) lo++;
               while (hi > 1 && !(tp[(hi - 1) * N + (j - 1)] > (float)LSMALL)) hi--;
               if (lo > hi) { lo = 1; hi = N - 1; }
               int arg = lo;
               double best = lk[lo] + (double)tp[(lo - 1) * N + (j - 1)];
               for (int i = lo + 1; i <= hi; i++) {
                  const double c = lk[i] + (double)tp[(i - 1) * N + (j - 1)];
                  if (c > best) { best = c; arg = i; }
               }
               gpre[(size_t)(t - 1) * nSlots + ms0 + j - 2] = best;
               gbp[(size_t)(t - 1) * nSlots + ms0 + j - 2] = (signed char)arg;
               nw[j] = LZERO;
               if (best > genThresh) {
                  nw[j] = best + (double)a.outp[ud.outp0 + (size_t)(ms0 + j - 2) * T + (t - 1)];
                  if (nw[j] > mx) mx = nw[j];
               }
            }
            lk[1] = LZERO;
            for (int j = 2; j < N; j++) lk[j] = nw[j];
            instMax[q] = mx;
            if (mx > myMax) myMax = mx;
            int lo = 2, hi = N - 1;
            while (lo < N && !(tp[(lo - 1) * N + (N - 1)] > (float)LSMALL)) lo++;
            while (hi > 1 && !(tp[(hi - 1) * N + (N - 1)] > (float)LSMALL)) hi--;
            if (lo > hi) { lo = 2; hi = N - 1; }
            int arg = lo;
            double best = lk[lo] + (double)tp[(lo - 1) * N + (N - 1)];
            for (int i = lo + 1; i <= hi; i++) {
               const double c = lk[i] + (double)tp[(i - 1) * N + (N - 1)];
               if (c > best) { best = c; arg = i; }
            }
            if (best > LSMALL) { exitL[q] = best; exArgS[q] = (signed char)arg; } else { exitL[q] = LZERO; exArgS[q] = 0; }
         }
         for (int o = 32; o > 0; o >>= 1) myMax = fmax(myMax, __shfl_xor(myMax, o));
         __syncthreads();
         if ((tid & 63) == 0) red[tid >> 6] = myMax;
         __syncthreads();
         if (tid == 0) {
            double g = red[0];
            for (int k = 1; k < VG_THREADS / 64; k++) g = fmax(g, red[k]);
            float th = (float)(g - (double)a.genBeam);
            if (th < (float)LSMALL) th = (float)LSMALL;
            gthr = th;
         }
         __syncthreads();
      }
      if (tid == 0) {                                     // pass 2 in chain order (HRec.c:2007-2016)
         const float genThresh = gthr;
         double carry = (t == 0) ? 0.0 : LZERO;
         bool haveCarry = (t == 0);
         for (int q = 0; q < Q; q++) {
            const int N = a.mN[ud.q0 + q];
            const float *tp = a.transP + a.mTp[ud.q0 + q];
            double *lk = like + (size_t)q * LS;
            if (haveCarry && carry > genThresh) {          // SetEntryState
               if (!active[q]) { active[q] = 1; instMax[q] = LZERO; exitL[q] = LZERO; exArgS[q] = 0; for (int i = 1; i < N; i++) lk[i] = LZERO; }
               if (carry > lk[1]) lk[1] = carry;
               if (lk[1] > instMax[q]) instMax[q] = lk[1];
            }
            haveCarry = false; carry = LZERO;
            if (active[q]) {
               if (instMax[q] < genThresh) {              // DetachInst
                  active[q] = 0; exitL[q] = LZERO; exArgS[q] = 0;
                  for (int i = 1; i < N; i++) lk[i] = LZERO;
               } else {
                  const float a1N = tp[N - 1];
                  if (a1N > (float)LSMALL) {              // StepHMM2
                     const double c = lk[1] + (double)a1N;
                     if (c > exitL[q]) { exitL[q] = c; exArgS[q] = 1; }
                  }
                  if (exitL[q] > genThresh) { carry = exitL[q]; haveCarry = true; }
               }
            }
            gent[(size_t)t * Q + q] = active[q] ? lk[1] : LZERO;
            gexl[(size_t)t * Q + q] = active[q] ? exitL[q] : LZERO;
            gexbp[(size_t)t * Q + q] = exArgS[q];
         }
         if (t == T) finalLike = (haveCarry && carry > LSMALL) ? carry : LZERO;
      }
      __syncthreads();
   }
   if (tid == 0) {
      a.total[u] = finalLike;
      a.status[u] = (finalLike > LSMALL) ? HTKAMD_UTT_OK : HTKAMD_UTT_SKIPPED;
   }
}

// one lane per utterance: walk the back-pointers (LatFromPaths / TranscriptionFromLattice restated on the trellis)
__global__ void k_viterbi_trace(VitArgs a)
{
   const int u = blockIdx.x * blockDim.x + threadIdx.x;
   if (u >= a.nUtt) return;
   const VitUtt ud = a.utt[u];
   const int T = ud.T, Q = ud.Q, nSlots = ud.nSlots;
   for (int k = 0; k < nSlots; k++) { a.segStart[ud.seg0 + k] = -1; a.segEnd[ud.seg0 + k] = -1; a.segScore[ud.seg0 + k] = 0.0; }
   for (int q = 0; q < Q; q++) { a.modStart[ud.mod0 + q] = -1; a.modEnd[ud.mod0 + q] = -1; a.modScore[ud.mod0 + q] = 0.0; }
   if (a.status[u] != HTKAMD_UTT_OK) return;
   const signed char *bp = a.bp + ud.tr0, *exbp = a.exbp + ud.mt0;
   const double *pre = a.pre + ud.tr0, *exl = a.exl + ud.mt0, *ent = a.entAt + ud.mt0;
   int tcur = T, q = Q;
   while (q >= 1) {
      const int slot0 = a.mSlot0[ud.q0 + q - 1];
      int st = exbp[(size_t)tcur * Q + q - 1];
      const double exitLike = exl[(size_t)tcur * Q + q - 1];
      double nextLike = exitLike;
      int segEndT = tcur;
      if (st == 1) {                                      // tee pass-through: the model takes no frame
         a.modStart[ud.mod0 + q - 1] = tcur; a.modEnd[ud.mod0 + q - 1] = tcur;
         a.modScore[ud.mod0 + q - 1] = exitLike - ent[(size_t)tcur * Q + q - 1];
         q--;
         continue;
      }
      a.modEnd[ud.mod0 + q - 1] = tcur;
      for (;;) {
         const size_t ix = (size_t)(tcur - 1) * nSlots + slot0 + st - 2;
         const int p = bp[ix];
         if (p != st) {
            a.segStart[ud.seg0 + slot0 + st - 2] = tcur - 1;
            a.segEnd[ud.seg0 + slot0 + st - 2] = segEndT;
            a.segScore[ud.seg0 + slot0 + st - 2] = nextLike - pre[ix];
            nextLike = pre[ix];
            segEndT = tcur - 1;
            if (p == 1) {
               a.modStart[ud.mod0 + q - 1] = tcur - 1;
               a.modScore[ud.mod0 + q - 1] = exitLike - ent[(size_t)(tcur - 1) * Q + q - 1];
               tcur--; q--;
               break;
            }
            st = p;
         }
         tcur--;
      }
   }
}

// ------------------------------------------------------------------------------------ host side
struct VBuf {
   void *p = nullptr; size_t cap = 0;
   int reserve(size_t bytes)
   {
      if (bytes <= cap) return HTKAMD_OK;
      if (p) (void)hipFree(p);
      p = nullptr; cap = 0;
      const size_t want = bytes + bytes / 8 + 64;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) { htkamd_set_error("viterbi: hipMalloc(%zu): %s", want, hipGetErrorString(e)); return HTKAMD_ENOMEM; }
      cap = want;
      return HTKAMD_OK;
   }
   void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct htkamd_viterbi {
   htkamd_model *m;
   int nUtt;
   std::vector<VitUtt> utt;
   std::vector<int> mN, mTp, mSlot0, slotState;
   std::vector<ScoreTask> tasks;
   size_t segTotal, modTotal;
   VBuf d_uttList;
   VBuf d_utt, d_mN, d_mTp, d_mSlot0, d_slotState, d_tasks, d_counter, d_outp, d_bp, d_pre, d_exl, d_ent, d_exbp;
   VBuf d_segStart, d_segEnd, d_segScore, d_modStart, d_modEnd, d_modScore, d_total, d_status;
};

extern "C" int htkamd_viterbi_create(htkamd_model *m, htkamd_viterbi **out)
{
   if (!m || !out) { htkamd_set_error("viterbi_create: NULL argument"); return HTKAMD_EINVAL; }
   if (m->maxN > VG_MAXN) { htkamd_set_error("viterbi_create: models with %d states; this path handles up to %d", m->maxN, VG_MAXN); return HTKAMD_EMODEL; }
   htkamd_viterbi *v = new htkamd_viterbi();
   v->m = m; v->nUtt = 0; v->segTotal = v->modTotal = 0;
   *out = v;
   return HTKAMD_OK;
}

extern "C" void htkamd_viterbi_destroy(htkamd_viterbi *v)
{
   if (!v) return;
   VBuf *all[] = {&v->d_uttList, &v->d_utt, &v->d_mN, &v->d_mTp, &v->d_mSlot0, &v->d_slotState, &v->d_tasks, &v->d_counter, &v->d_outp, &v->d_bp,
                  &v->d_pre, &v->d_exl, &v->d_ent, &v->d_exbp, &v->d_segStart, &v->d_segEnd, &v->d_segScore, &v->d_modStart,
                  &v->d_modEnd, &v->d_modScore, &v->d_total, &v->d_status};
   for (VBuf *b : all) b->release();
   delete v;
}

template <typename T> static int vupload(VBuf &b, const std::vector<T> &v, hipStream_t s)
{
   int rc = b.reserve(sizeof(T) * (v.size() ? v.size() : 1));
   if (rc) return rc;
   if (!v.empty()) HIPCHECK(hipMemcpyAsync(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s));
   return HTKAMD_OK;
}

extern "C" int htkamd_viterbi_align(htkamd_viterbi *v, const htkamd_batch_desc *b, float genBeam, void *stream)
{
   return htkamd_viterbi_align_mode(v, b, genBeam, HTKAMD_SCORE_EXACT, stream);
}

extern "C" int htkamd_viterbi_align_mode(htkamd_viterbi *v, const htkamd_batch_desc *b, float genBeam, int scoreMode, void *stream)
{
   if (scoreMode & ~(HTKAMD_SCORE_SOUTP | HTKAMD_SCORE_DIAGC)) { htkamd_set_error("viterbi_align: score mode %d (0, HTKAMD_SCORE_SOUTP, HTKAMD_SCORE_DIAGC)", scoreMode); return HTKAMD_EINVAL; }
   if (!v || !b || b->nUtt < 0 || (b->nUtt > 0 && (!b->dX || !b->frameOff || !b->labOff || !b->labs))) {
      htkamd_set_error("viterbi_align: bad argument"); return HTKAMD_EINVAL;
   }
   const htkamd_model *m = v->m;
   hipStream_t s = (hipStream_t)stream;
   const int U = b->nUtt;
   v->nUtt = U;
   v->utt.assign(U, VitUtt());
   v->mN.clear(); v->mTp.clear(); v->mSlot0.clear(); v->slotState.clear(); v->tasks.clear();
   size_t outp = 0, tr = 0, mt = 0, seg = 0, mod = 0;
   // chains of up to 64 / 128 / 256 / 512 models of up to VMAXN states: 1 / 2 / 4 / 8 wavefronts per utterance (k_viterbi_w);
   // longer chains or bigger models: a workgroup per utterance (k_viterbi_g)
   std::vector<int> cls[5];
   int maxQ = 1;
   for (int u = 0; u < U; u++) {
      VitUtt &d = v->utt[u];
      const int T = b->frameOff[u + 1] - b->frameOff[u], Q = b->labOff[u + 1] - b->labOff[u];
      const int *labs = b->labs + b->labOff[u];
      d.T = T; d.Q = Q; d.frame0 = b->frameOff[u]; d.q0 = (int)v->mN.size(); d.slot0 = (int)v->slotState.size();
      d.status = HTKAMD_UTT_OK; d.pad = 0; d.outp0 = outp; d.tr0 = tr; d.mt0 = mt; d.seg0 = seg; d.mod0 = mod;
      int nSlots = 0;
      if (T <= 0 || Q <= 0) { d.status = HTKAMD_UTT_SKIPPED; d.nSlots = 0; cls[0].push_back(u); continue; }   // the kernel reports it
      if (Q > 4000) { htkamd_set_error("viterbi_align: utterance %d has %d models (max 4000)", u, Q); return HTKAMD_EINVAL; }
      {
         const int c = (m->maxN > VMAXN) ? 4 : (Q <= 64 ? 0 : Q <= 128 ? 1 : Q <= 256 ? 2 : Q <= 512 ? 3 : 4);
         cls[c].push_back(u);
         if (c == 4 && Q > maxQ) maxQ = Q;
      }
      for (int q = 1; q <= Q; q++) {
         const int h = labs[q - 1];
         if (h < 0 || h >= m->H) { htkamd_set_error("viterbi_align: utterance %d label %d: HMM index %d out of range", u, q, h); return HTKAMD_EINVAL; }
         const int ti = m->h_hmmTrans[h], N = m->h_transN[ti];
         v->mN.push_back(N); v->mTp.push_back(m->h_transOff[ti]); v->mSlot0.push_back(nSlots);
         for (int j = 2; j < N; j++) v->slotState.push_back(m->h_hmmState[m->h_hmmStateOff[h] + (j - 2) * m->NSt]);      // several streams: the state's first element (ScoreArgs::NSt)
         nSlots += N - 2;
      }
      d.nSlots = nSlots;
      for (int t0 = 0; t0 < T; t0 += SCORE_TILE_FRAMES)
         for (int k0 = 0; k0 < nSlots; k0 += SCORE_TASK_SLOTS) {
            ScoreTask tk;
            tk.frame0 = d.frame0 + t0; tk.nFrames = (T - t0 < SCORE_TILE_FRAMES) ? T - t0 : SCORE_TILE_FRAMES;
            tk.slot0 = d.slot0 + k0; tk.nSlots = (nSlots - k0 < SCORE_TASK_SLOTS) ? nSlots - k0 : SCORE_TASK_SLOTS;
            tk.outSlot0 = k0; tk.ldo = T; tk.outBase = d.outp0 + (size_t)t0;
            v->tasks.push_back(tk);
         }
      outp += (size_t)T * nSlots; tr += (size_t)T * nSlots; mt += (size_t)(T + 1) * Q; seg += nSlots; mod += Q;
   }
   v->segTotal = seg; v->modTotal = mod;
   int clsOff[6] = {0, 0, 0, 0, 0, 0};
   std::vector<int> uttList;
   for (int c = 0; c < 5; c++) { uttList.insert(uttList.end(), cls[c].begin(), cls[c].end()); clsOff[c + 1] = (int)uttList.size(); }
   if (uttList.empty()) uttList.push_back(0);
   int rc;
   if ((rc = vupload(v->d_uttList, uttList, s))) return rc;
   if ((rc = vupload(v->d_utt, v->utt, s)) || (rc = vupload(v->d_mN, v->mN, s)) || (rc = vupload(v->d_mTp, v->mTp, s)) ||
       (rc = vupload(v->d_mSlot0, v->mSlot0, s)) || (rc = vupload(v->d_slotState, v->slotState, s)) || (rc = vupload(v->d_tasks, v->tasks, s)))
      return rc;
   auto nz = [](size_t x) { return x ? x : (size_t)1; };
   if ((rc = v->d_counter.reserve(64)) || (rc = v->d_outp.reserve(4 * nz(outp))) || (rc = v->d_bp.reserve(nz(tr))) ||
       (rc = v->d_pre.reserve(8 * nz(tr))) || (rc = v->d_exl.reserve(8 * nz(mt))) || (rc = v->d_ent.reserve(8 * nz(mt))) ||
       (rc = v->d_exbp.reserve(nz(mt))) || (rc = v->d_segStart.reserve(4 * nz(seg))) || (rc = v->d_segEnd.reserve(4 * nz(seg))) ||
       (rc = v->d_segScore.reserve(8 * nz(seg))) || (rc = v->d_modStart.reserve(4 * nz(mod))) || (rc = v->d_modEnd.reserve(4 * nz(mod))) ||
       (rc = v->d_modScore.reserve(8 * nz(mod))) || (rc = v->d_total.reserve(8 * nz(U))) || (rc = v->d_status.reserve(4 * nz(U))))
      return rc;
   if (U == 0) return HTKAMD_OK;

   ScoreArgs sa;
   sa.tasks = (const ScoreTask *)v->d_tasks.p; sa.nTasks = (int)v->tasks.size(); sa.X = b->dX;
   sa.slotState = (const int *)v->d_slotState.p; sa.out = (float *)v->d_outp.p;
   sa.stateCompOff = m->d_stateCompOff; sa.compGauss = m->d_compGauss; sa.compLogWt = m->d_compLogWt;
   sa.gparam = m->d_gparam; sa.PS = m->PS; sa.D = m->D; sa.minLogExp = m->minLogExp;
   sa.laddTab = m->d_laddTab; sa.taskCounter = (int *)v->d_counter.p;
   if (scoreMode & HTKAMD_SCORE_DIAGC) { if ((rc = htkamd_model_device_tables((htkamd_model *)m))) return rc; }
   sa.var = m->d_var;
   sa.NSt = m->NSt; sa.streamWt = m->d_streamWt;
   if (m->tiedMix) {                                      // hsKind TIEDHS: PrecomputeTMix(tmBeam) per frame + SOutP's pool sum (HRec.c:1987, 493-503)
      if ((rc = htkamd_tm_score_block(m, sa, b->frameOff[U], m->tmBeam, s))) return rc;
   } else
   if ((rc = htkamd_launch_score_exact(m, sa, s, nullptr, nullptr, (scoreMode & HTKAMD_SCORE_SOUTP) != 0, (scoreMode & HTKAMD_SCORE_DIAGC) != 0))) return rc;

   VitArgs va;
   va.utt = (const VitUtt *)v->d_utt.p; va.nUtt = U;
   va.mN = (const int *)v->d_mN.p; va.mTp = (const int *)v->d_mTp.p; va.mSlot0 = (const int *)v->d_mSlot0.p;
   va.transP = m->d_transP; va.outp = (const float *)v->d_outp.p;
   va.bp = (signed char *)v->d_bp.p; va.pre = (double *)v->d_pre.p;
   va.exl = (double *)v->d_exl.p; va.entAt = (double *)v->d_ent.p; va.exbp = (signed char *)v->d_exbp.p;
   va.segStart = (int *)v->d_segStart.p; va.segEnd = (int *)v->d_segEnd.p; va.segScore = (double *)v->d_segScore.p;
   va.modStart = (int *)v->d_modStart.p; va.modEnd = (int *)v->d_modEnd.p; va.modScore = (double *)v->d_modScore.p;
   va.total = (double *)v->d_total.p; va.status = (int *)v->d_status.p;
   va.genBeam = genBeam;
   for (int c = 3; c >= 0; c--) {                        // the longest chains first
      va.uttList = (const int *)v->d_uttList.p + clsOff[c]; va.nList = clsOff[c + 1] - clsOff[c];
      if (va.nList <= 0) continue;
      if (c == 0) hipLaunchKernelGGL((k_viterbi_w<VMAXN, 1>), dim3((va.nList + 3) / 4), dim3(256), 0, s, va);
      else if (c == 1) hipLaunchKernelGGL((k_viterbi_w<VMAXN, 2>), dim3(va.nList), dim3(128), 0, s, va);
      else if (c == 2) hipLaunchKernelGGL((k_viterbi_w<VMAXN, 4>), dim3(va.nList), dim3(256), 0, s, va);
      else hipLaunchKernelGGL((k_viterbi_w<VMAXN, 8>), dim3(va.nList), dim3(512), 0, s, va);
      HIPCHECK(hipGetLastError());
   }
   va.uttList = (const int *)v->d_uttList.p + clsOff[4]; va.nList = clsOff[5] - clsOff[4];
   if (va.nList > 0) {
      const size_t lds = sizeof(double) * ((size_t)maxQ * (m->maxN + 1) + 2 * (size_t)maxQ) + 2 * (size_t)maxQ + 64;
      if (lds > 150 * 1024) { htkamd_set_error("viterbi_align: %d models of up to %d states need %zu bytes of LDS", maxQ, m->maxN, lds); return HTKAMD_EMODEL; }
      if (lds > 64 * 1024) HIPCHECK(hipFuncSetAttribute((const void *)k_viterbi_g, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_viterbi_g, dim3(va.nList), dim3(VG_THREADS), lds, s, va);
   }
   HIPCHECK(hipGetLastError());
   hipLaunchKernelGGL(k_viterbi_trace, dim3((U + 63) / 64), dim3(64), 0, s, va);
   HIPCHECK(hipGetLastError());
   // the host tables must outlive the async uploads
   HIPCHECK(hipStreamSynchronize(s));
   return HTKAMD_OK;
}

extern "C" int htkamd_viterbi_sizes(const htkamd_viterbi *v, size_t *nSeg, size_t *nMod)
{
   if (!v) { htkamd_set_error("viterbi_sizes: NULL"); return HTKAMD_EINVAL; }
   if (nSeg) *nSeg = v->segTotal;
   if (nMod) *nMod = v->modTotal;
   return HTKAMD_OK;
}

extern "C" int htkamd_viterbi_results(htkamd_viterbi *v, int *segStart, int *segEnd, double *segScore,
                                      int *modStart, int *modEnd, double *modScore, double *total, int *status, void *stream)
{
   if (!v) { htkamd_set_error("viterbi_results: NULL"); return HTKAMD_EINVAL; }
   hipStream_t s = (hipStream_t)stream;
   HIPCHECK(hipStreamSynchronize(s));
   if (v->nUtt == 0) return HTKAMD_OK;
   if (segStart) HIPCHECK(hipMemcpy(segStart, v->d_segStart.p, 4 * v->segTotal, hipMemcpyDeviceToHost));
   if (segEnd) HIPCHECK(hipMemcpy(segEnd, v->d_segEnd.p, 4 * v->segTotal, hipMemcpyDeviceToHost));
   if (segScore) HIPCHECK(hipMemcpy(segScore, v->d_segScore.p, 8 * v->segTotal, hipMemcpyDeviceToHost));
   if (modStart) HIPCHECK(hipMemcpy(modStart, v->d_modStart.p, 4 * v->modTotal, hipMemcpyDeviceToHost));
   if (modEnd) HIPCHECK(hipMemcpy(modEnd, v->d_modEnd.p, 4 * v->modTotal, hipMemcpyDeviceToHost));
   if (modScore) HIPCHECK(hipMemcpy(modScore, v->d_modScore.p, 8 * v->modTotal, hipMemcpyDeviceToHost));
   if (total) HIPCHECK(hipMemcpy(total, v->d_total.p, 8 * (size_t)v->nUtt, hipMemcpyDeviceToHost));
   if (status) HIPCHECK(hipMemcpy(status, v->d_status.p, 4 * (size_t)v->nUtt, hipMemcpyDeviceToHost));
   return HTKAMD_OK;
}
