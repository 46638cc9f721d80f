// decode.hip -- K7: 1-best Viterbi decoding over a recognition network (HVite -w), one workgroup per utterance.
//
// Reference semantics (HRec.c with nToks = 1, no alignment records; oracle/orc_decode.c is the restatement this kernel
// is tested against):
//   StartRecognition :1884 / ProcessObservation :1935-2030 -- pass 1 StepHMM1 (:642) on every model instance with the
//   PREVIOUS frame's genThresh, beam tops genMaxTok / wordMaxTok (exit + LikeToWord :1172); thresholds as floats floored at
//   LSMALL; pass 2: instances under genThresh are detached, StepInst2 (:1360) = StepWord2 (:1046: word penalty, pron prob,
//   Path record) / StepHMM2 (:790, tee models), wordThresh on word tokens, tokens above genThresh go down every link with
//   like += lm*scale into SetEntryState (:1303, strict >).  CompleteRecognition :2054 + LatFromPaths :1512 +
//   TranscriptionFromLattice :2176 give the word labels and their LArcTotLike scores.
//
// MI355X mapping.  Output probabilities are NOT evaluated inside the token loop: K1 (exact mode) scores every tied state the
// network uses for all frames first -- the dense, compute-bound form of the problem -- so the sequential part only reads
// floats.  The token loop is latency-bound (500 dependent frames), so each utterance gets one 1024-thread workgroup and
// the machine is filled with utterances.  The reference PUSHES tokens along links in list order; here every node PULLS:
//   phase A   thread per model node: entry token = best over predecessors' exit tokens (reverse CSR), then StepHMM1 on
//             registers; block-wide max for the two beam tops; thread 0 publishes the thresholds.
//   phase L   the zero-time nodes (word ends, null nodes, tee models) level by level in topological order, thread per node
//             pulling from its predecessors; nodes with a large fan-in (the loop/back-off null nodes: thousands of word
//             ends) are reduced by the whole workgroup.  One barrier per level.
// Detaching an instance only changes which tokens are visible later, so it is folded into the next frame's phase A
// (instance max of the previous frame against the threshold of the previous frame).  Word-end Path records go to a dense
// [frame][word node] table (no allocation, no garbage collection); the final token's chain is walked by one thread.
// Tokens are (double like, float lm, int path), kept as separate arrays in global memory (L2-resident per utterance).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"

#include "decode.h"

__device__ __forceinline__ Tok null_tok() { Tok t; t.like = LZERO; t.lm = 0.0f; t.path = -1; return t; }

// entry token of node n: best over predecessors (SetEntryState over StepInst2's sends), first maximum wins.
// *tie is set when a second token of EXACTLY the winner's likelihood and another history (path or LM share) was met: which of the two
// the reference keeps depends on the order of its instance list (decode_ord.hip decodes such utterances again, in that order).
struct PullAcc { Tok best; int arg; bool tb; };
__device__ __forceinline__ PullAcc pull_start() { PullAcc p; p.best = null_tok(); p.arg = 0x7fffffff; p.tb = false; return p; }
// one predecessor: its exit token e, the link (ps: node | bit 31 for a word / null predecessor; lm), its position k in the list
__device__ __forceinline__ void pull_fold(PullAcc &p, const DecArgs &a, const Tok e, const int ps, const float lm, const int k, const float gT, const float wT)
{
   if (!(e.like > gT)) return;
   if (ps < 0 && e.like < wT) return;                                  // word-end beam on word/null tokens
   const double c = e.like + lm * a.lmScale;
   if (!(c > gT)) return;
   if (c > p.best.like) { p.best.like = c; p.best.lm = e.lm + lm; p.best.path = e.path; p.arg = k; p.tb = false; }
   else if (c == p.best.like && (e.path != p.best.path || e.lm + lm != p.best.lm)) p.tb = true;
}
// The list is walked four predecessors at a time: their link words first, then their tokens, then the comparisons in list order.  One by
// one, each token's load waited for its link's load and the next link for the comparison before it: two memory latencies per predecessor,
// 68 % of k_decode's cycles on the 6 000-word bigram network (tools/dec_diag.py with -DDEC_CLK).
// EXL: predecessors with a copy in LDS (DecNet::zl) are taken from there: wl the WORD nodes' likelihoods, nullL the null nodes' tokens, tNW = frame * nWordNodes
template <bool EXL = false>
__device__ __forceinline__ Tok pull_range(const DecArgs &a, const Tok *ex, int k0, int k1, int kstep, float gT, float wT, int *argk, bool *tie,
                                          const double *wl = nullptr, const Tok *nullL = nullptr, const int tNW = 0)
{
   PullAcc p = pull_start();
   for (int k = k0; k < k1; k += 4 * kstep) {
      int2 r[4]; Tok e[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { const int kk = k + i * kstep; r[i] = (EXL ? a.net.predRecL : a.net.predRec)[kk < k1 ? kk : k0]; }
#pragma unroll
      for (int i = 0; i < 4; i++) {
         const int x = r[i].x;
         if (EXL && (x & 0x40000000)) {
            const int wi = x & 0x1fffffff;
            const double l = wl[wi];
            e[i].like = l; e[i].lm = 0.0f; e[i].path = (l > LSMALL) ? tNW + wi : -1;
         } else if (EXL && (x & 0x20000000)) e[i] = nullL[x & 0x1fffffff];
         else e[i] = ex[x & 0x1fffffff];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) { const int kk = k + i * kstep; if (kk < k1) pull_fold(p, a, e[i], r[i].x, __int_as_float(r[i].y), kk, gT, wT); }
   }
   *argk = p.arg;
   if (p.tb) *tie = true;
   return p.best;
}

// the two beam tops in one pass over the workgroup
template <int NTHR>
__device__ __forceinline__ void block_max2(double &v, double &w, double *red, double *red2)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) { const double x = __shfl_xor(v, o), y = __shfl_xor(w, o); v = (x > v) ? x : v; w = (y > w) ? y : w; }
   const int wv = threadIdx.x >> 6;
   __syncthreads();
   if ((threadIdx.x & 63) == 0) { red[wv] = v; red2[wv] = w; }
   __syncthreads();
   v = red[0]; w = red2[0];
   for (int i = 1; i < NTHR / 64; i++) { v = (red[i] > v) ? red[i] : v; w = (red2[i] > w) ? red2[i] : w; }
}

template <int NTHR>
__device__ __forceinline__ double block_max(double v, double *red)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = (w > v) ? w : v; }
   const int wv = threadIdx.x >> 6;
   __syncthreads();
   if ((threadIdx.x & 63) == 0) red[wv] = v;
   __syncthreads();
   double r = red[0];
   for (int i = 1; i < NTHR / 64; i++) r = (red[i] > r) ? red[i] : r;
   return r;
}

#define DEC_LDS_TP 4096            /* floats of transition matrices cached in LDS (all of them, else global memory) */
#ifdef DEC_NEED                    /* diagnostic build: distinct tied states whose score a live token asked for, per frame (tools/dec_diag.py) */
__shared__ unsigned int needB[256];            // (up to 8 192 score slots)
#define DEC_NEED_MARK(slot_) atomicOr(&needB[((slot_) >> 5) & 255], 1u << ((slot_) & 31))
#else
#define DEC_NEED_MARK(slot_) do { } while (0)
#endif

// StepHMM1 (HRec.c:642) on one model instance: s[1 .. NS-1] = the state tokens on entry (s[1] the entry token) and the new ones on
// return (s[1] null: the entry is consumed); exT = the exit token, mx = the instance's maximum, wordTop raised by exit + LikeToWord.
template <int MX, bool SLOTS = false>
__device__ __forceinline__ void hmm_step1(const DecArgs &a, const DecUtt &ud, const int4 ni, const float *tp, Tok (&s)[MX], const float gT, const int t,
                                          const float wdlk, Tok &exT, double &mx, double &wordTop, const int sl2 = 0, const int sl3 = 0, const int sl4 = 0, const int se = 0)
{
   const DecNet &N = a.net;
   const int NS = (ni.x >> 4) & 255;
   Tok nw[MX];
   if constexpr (SLOTS && MX == 5) {
      // The usual model -- three emitting states, no skips: ranges (1,2) (2,3) (3,4), exit (4,4) -- when every live lane of the wavefront has
      // it: the same comparisons as below for those ranges, without the twelve guarded (state, predecessor) combinations of the general form
      constexpr int SE_LTR5 = 1 | (2 << 3) | (2 << 6) | (3 << 9) | (3 << 12) | (4 << 15) | (4 << 18) | (4 << 21);
      if (__builtin_amdgcn_ballot_w64(!(NS == 5 && se == SE_LTR5)) == 0ull) {
         const float t12 = tp[1], t22 = tp[6], t23 = tp[7], t33 = tp[12], t34 = tp[13], t44 = tp[18], t45 = tp[19];
         const float tIn[3] = {t12, t23, t34}, tSelf[3] = {t22, t33, t44};
         const int slj[3] = {sl2, sl3, sl4};
#pragma unroll
         for (int j = 2; j < 5; j++) {
            nw[j] = null_tok();
            const double c1 = s[j - 1].like + tIn[j - 2], c2 = s[j].like + tSelf[j - 2];
            Tok best = s[j - 1]; double bl = c1;
            if (c2 > bl) { best = s[j]; bl = c2; }
            best.like = bl;
            if (best.like > gT) {
               best.like += __builtin_nontemporal_load(a.score + ud.score0 + (size_t)(t - 1) * a.ns + slj[j - 2]);
               DEC_NEED_MARK(slj[j - 2]);
               nw[j] = best;
               if (best.like > mx) mx = best.like;
            }
         }
         Tok best = nw[4];
         best.like = nw[4].like + t45;
         if (best.like > LSMALL) {
            exT = best;
            const double w = best.like + wdlk;
            if (w > wordTop) wordTop = w;
         }
         s[1] = null_tok();
#pragma unroll
         for (int j = 2; j < 5; j++) s[j] = nw[j];
         return;
      }
   }
#pragma unroll
   for (int j = 2; j < MX; j++) {
      nw[j] = null_tok();
      if (j < NS) {
         // CreateSEIndex (HRec.c:1403): predecessor range with a transition, first maximum wins
         int lo = 1, hi = NS - 1;
         if constexpr (SLOTS) { lo = (se >> (6 * (j - 2))) & 7; hi = (se >> (6 * (j - 2) + 3)) & 7; }      // (the ranges of the register-resident models come with their records)
         else {
            while (lo < NS && !(tp[(lo - 1) * NS + (j - 1)] > LSMALL)) lo++;
            while (hi > 1 && !(tp[(hi - 1) * NS + (j - 1)] > LSMALL)) hi--;
            if (lo > hi) { lo = 1; hi = NS - 1; }
         }
         Tok best = s[1]; double bl = LZERO;
#pragma unroll
         for (int i = 1; i < MX; i++)
            if (i >= lo && i <= hi) {
               const double c = s[i].like + tp[(i - 1) * NS + (j - 1)];
               if (i == lo || c > bl) { best = s[i]; bl = c; }
            }
         best.like = bl;
         if (best.like > gT) {
            int slot;
            if constexpr (SLOTS) slot = (j == 2) ? sl2 : (j == 3) ? sl3 : sl4;      // (the register-resident models carry their score slots)
            else slot = N.stateSlot[N.hmmState[ni.w + (j - 2)]];
            best.like += __builtin_nontemporal_load(a.score + ud.score0 + (size_t)(t - 1) * a.ns + slot);      // (a frame's column is read once)
            DEC_NEED_MARK(slot);
            nw[j] = best;
            if (best.like > mx) mx = best.like;
         }
      }
   }
   {
      int lo = 2, hi = NS - 1;
      if constexpr (SLOTS) { lo = (se >> 18) & 7; hi = (se >> 21) & 7; }
      else {
         while (lo < NS && !(tp[(lo - 1) * NS + (NS - 1)] > LSMALL)) lo++;
         while (hi > 1 && !(tp[(hi - 1) * NS + (NS - 1)] > LSMALL)) hi--;
         if (lo > hi) { lo = 2; hi = NS - 1; }
      }
      Tok best = nw[2]; double bl = LZERO;
#pragma unroll
      for (int i = 2; i < MX; i++)
         if (i >= lo && i <= hi) {
            const double c = nw[i].like + tp[(i - 1) * NS + (NS - 1)];
            if (i == lo || c > bl) { best = nw[i]; bl = c; }
         }
      best.like = bl;
      if (best.like > LSMALL) {
         exT = best;
         const double w = best.like + wdlk;
         if (w > wordTop) wordTop = w;
      }
   }
   s[1] = null_tok();                                      // entry consumed
#pragma unroll
   for (int j = 2; j < MX; j++) if (j < NS) s[j] = nw[j];
}

// StepWord2 (HRec.c:1046) for the token `st` that entered word node n at frame t: the Path record, the exit token
__device__ __forceinline__ Tok word_step2(const DecArgs &a, const DecUtt &ud, const int n, const int t, const Tok st)
{
   const DecNet &N = a.net;
   Tok e = st;
   e.like += a.wordPen;
   e.like += N.pronProb[n] * a.prScale;
   const size_t pid = (size_t)t * N.nWordNodes + N.wordIdx[n];
   __builtin_nontemporal_store(st.path, a.pathPrev + ud.path0 + pid); __builtin_nontemporal_store(e.like, a.pathLike + ud.path0 + pid);
   __builtin_nontemporal_store(e.lm, a.pathLm + ud.path0 + pid);
   e.path = (int)pid; e.lm = 0.0f;
   return e;
}

// k_decode<NPT, NTHR>.  NPT = 0, NTHR = 1024: every token in global memory (any network).  NPT > 0 (round 4): the first NPT * NTHR model
// nodes of N.hmmNodes -- plain models of at most three emitting states, the host put them first (DecNet::nReg) -- keep their state
// tokens, their entry token and, between pass 1 and pass 2, their exit token in REGISTERS of the thread that owns them for the whole
// utterance (node hmmNodes[tid + k NTHR] is thread tid's k-th: 16 VGPRs per node); a word / null node whose ONLY predecessor is such a
// model is stepped by that thread from the register (DecNet::regFused), and a model all of whose successors are stepped that way never
// writes its exit token to memory.  Round 3 moved 2.2 MB per utterance and frame on the 6 000-word loop -- 277 GB per launch of 256
// utterances, 563 MB of live state against 256 MB of Infinity Cache; what is left in memory here: exit tokens of the word nodes, instance
// maxima, Path records, the score column.
#define DEC_MAXR 5                 /* states of a register-resident model incl. entry / exit */
template <int NPT, int NTHR, bool HASG, bool EXL = false>       // HASG: model nodes outside the registers exist (tee models, more than three emitting states, overflow); EXL: see DecNet::zl
__global__ __launch_bounds__(NTHR) void k_decode(DecArgs a)
{
   __shared__ double red[NTHR / 64];
   __shared__ double red2[NTHR / 64];
   __shared__ int redk[NTHR / 64];
   __shared__ float thr[2];
   __shared__ float ltpS[EXL ? 1 : DEC_LDS_TP];    // (EXL: the matrices behind the dynamic block, as many floats as there are)
   __shared__ Tok nullL[EXL ? DEC_NULL_LDS : 1];
#ifdef DEC_NEED
   unsigned long long needSum = 0;
#endif
   __shared__ int uhist[256];
   __shared__ unsigned int usel[4];            // -u: [0] attached instances, [1] key prefix, [2] rank still to skip, [3] scratch
   const int u = blockIdx.x, tid = threadIdx.x;
   if (u >= a.nUtt) return;
   const DecUtt ud = a.utt[u];
   const DecNet &N = a.net;
   const int T = ud.T;
   Tok *tok = a.tok + ud.tok0, *ex = a.ex + ud.node0;
   double *imax = a.imax + ud.node0;
   extern __shared__ Tok xs[];
   // dynamic LDS: xs [NPT NTHR] tokens | EXL: wl [nWordNodes] doubles, ltp [nTpFloats] floats | else: rmaxL [NPT NTHR] floats
   double *wl = (double *)(xs + (size_t)(NPT > 0 ? NPT : 0) * NTHR);
   float *ltp = EXL ? (float *)(wl + N.nWordNodes) : ltpS;
   const bool tpInLds = EXL || N.nTpFloats <= DEC_LDS_TP;     // (EXL is launched only when they fit)
   if (tpInLds) for (int i = tid; i < N.nTpFloats; i += NTHR) ltp[i] = N.transP[i];
   if constexpr (EXL) {
      for (int i = tid; i < N.nWordNodes; i += NTHR) wl[i] = LZERO;
      for (int i = tid; i < DEC_NULL_LDS; i += NTHR) nullL[i] = null_tok();
   }
   // the LDS copy of a zero-time node's exit token (every store to ex[] of such a node comes through here)
   // (a node with a slot is read from LDS by every pull of this kernel: its token stays out of memory -- 96 KB per utterance and frame on
   // the 6 000-word network, which with the Path records pushed the network's tables out of the L2 every frame)
   auto put_ex = [&](const int n, const Tok e, const int zl) {
      if constexpr (EXL) {
         if (zl >= 0) { if (zl & 0x40000000) wl[zl & 0x3fffffff] = e.like; else nullL[zl] = e; return; }
      }
      ex[n] = e;
   };
   const float *tpBase = tpInLds ? ltp : N.transP;
   bool tie = false;                           // this thread met two equally likely tokens with different histories (pull_range)
   unsigned int nLive = 0, nDead = 0;          // this thread's register-resident model steps with / without a live token
#ifdef DEC_CLK                                 // cycle stamps of thread 0 at the phase boundaries (tools/dec_diag.py): prune | models | beam tops | fused words | levels | entries
   unsigned long long clk[6] = {0, 0, 0, 0, 0, 0}, c0 = 0;
#define DEC_STAMP(i_) do { const unsigned long long c_ = __builtin_readcyclecounter(); clk[i_] += c_ - c0; c0 = c_; } while (0)
#else
#define DEC_STAMP(i_) do { } while (0)
#endif
   const int nReg = (NPT > 0) ? N.nReg : 0;    // model nodes hmmNodes[0 .. nReg) live in registers
   // this thread's register-resident models: rs[k][0 ..] = the tokens of states 2 .. ; the entry token (the exit token between the
   // passes) of its k-th model waits in LDS, xs[k NTHR + tid] (registers for it too made the compiler spill at 12 models per thread)
   Tok rs[NPT > 0 ? NPT : 1][DEC_MAXR - 2];
   float *rmaxL = (float *)(xs + (size_t)(NPT > 0 ? NPT : 0) * NTHR);      // NetInst.max of the register-resident models (a LogFloat), [k NTHR + tid], behind xs
   float rmaxR[(EXL && NPT > 0) ? NPT : 1];                                // ... in registers where LDS holds the word ends instead
#define rmax(k_) (*(EXL ? &rmaxR[k_] : &rmaxL[(k_) * NTHR + tid]))
   if constexpr (NPT > 0) {
#pragma unroll
      for (int k = 0; k < NPT; k++) {
         xs[k * NTHR + tid] = null_tok();
         rmax(k) = (float)LZERO;
#pragma unroll
         for (int i = 0; i < DEC_MAXR - 2; i++) rs[k][i] = null_tok();
      }
   }

   for (int i = tid; i < N.nTok; i += NTHR) tok[i] = null_tok();
   for (int i = tid; i < N.nNodes; i += NTHR) { ex[i] = null_tok(); imax[i] = LZERO; }
   if (tid == 0) { thr[0] = (float)LSMALL; thr[1] = (float)LSMALL; }
   __syncthreads();

#ifdef DEC_CLK
   c0 = __builtin_readcyclecounter();
#endif
   for (int t = 0; t <= T; t++) {
      if (t >= 1 && a.maxActive > 0) {
         // ---- maximum-model pruning (ProcessObservation HRec.c:1966-1985): when more than maxActive instances are attached, those
         // whose max (a float, NetInst.max) lies below the (maxActive+1)-th largest are detached before pass 1.  An instance is
         // attached when its max reached the previous frame's threshold (imax carries the pass-1 maximum raised by any token that
         // entered in pass 2; word / null nodes hold the token they were given).  Selection: radix select on the float keys.
         const float gTp = thr[0];
         if (tid == 0) usel[0] = 0;
         __syncthreads();
         int cnt = 0;
         for (int n = tid; n < N.nNodes; n += NTHR) { const double v = imax[n]; if (v >= gTp && v > LSMALL) cnt++; }
         if (cnt) atomicAdd(&usel[0], (unsigned)cnt);
         __syncthreads();
         const int nact = (int)usel[0];
         if (nact > a.maxActive) {
            if (tid == 0) { usel[1] = 0; usel[2] = (unsigned)a.maxActive; }
            unsigned int mask = 0;
            for (int pass = 0; pass < 4; pass++) {
               const int shift = 24 - 8 * pass;
               for (int i = tid; i < 256; i += NTHR) uhist[i] = 0;
               __syncthreads();
               const unsigned int prefix = usel[1];
               for (int n = tid; n < N.nNodes; n += NTHR) {
                  const double v = imax[n];
                  if (!(v >= gTp && v > LSMALL)) continue;
                  unsigned int k = __float_as_uint((float)v);
                  k ^= (k >> 31) ? 0xFFFFFFFFu : 0x80000000u;          // ascending order of the floats
                  if ((k & mask) == prefix) atomicAdd(&uhist[(k >> shift) & 255], 1);
               }
               __syncthreads();
               if (tid == 0) {
                  unsigned int skip = usel[2], cum = 0; int b = 255;
                  for (; b > 0; b--) { if (cum + (unsigned)uhist[b] > skip) break; cum += (unsigned)uhist[b]; }
                  usel[1] = prefix | ((unsigned)b << shift); usel[2] = skip - cum;
               }
               mask |= 255u << shift;
               __syncthreads();
            }
            unsigned int kk = usel[1];
            kk ^= (kk >> 31) ? 0x80000000u : 0xFFFFFFFFu;
            const float uth = __uint_as_float(kk);
            if (uth > (float)LSMALL) {
               if constexpr (NPT > 0) {                                // the tokens that live in registers: their owners drop them (before imax changes)
#pragma unroll
                  for (int k = 0; k < NPT; k++) {
                     const int hk = tid + k * NTHR;
                     if (hk < nReg) {
                        const double v = imax[N.hmmNodes[hk]];
                        if (v >= gTp && v > LSMALL && v < (double)uth) {
                           xs[k * NTHR + tid] = null_tok();
                           rmax(k) = (float)LZERO;
#pragma unroll
                           for (int i = 0; i < DEC_MAXR - 2; i++) rs[k][i] = null_tok();
                        }
                     }
                  }
                  __syncthreads();
               }
               for (int n = tid; n < N.nNodes; n += NTHR) {
                  const double v = imax[n];
                  if (!(v >= gTp && v > LSMALL) || !(v < (double)uth)) continue;
                  imax[n] = LZERO; put_ex(n, null_tok(), EXL ? N.zl[n] : -1);      // DetachInst: every token of the instance goes, the entry token too
                  const int4 ni = N.nodeInfo[n];
                  const int nt = ((ni.x & 15) == HTKAMD_NODE_HMM) ? ((ni.x >> 4) & 255) - 1 : 1;
                  for (int i = 0; i < nt; i++) tok[ni.y + i] = null_tok();
               }
            }
            __syncthreads();
         }
      }
      DEC_STAMP(0);
#ifdef DEC_NEED                                         /* -DDEC_NEED=1: per frame; -DDEC_NEED=16: the union over blocks of 16 frames (what ANY block-wise scheme has to score at least) */
      if ((t - 1) % (DEC_NEED + 0 > 0 ? DEC_NEED + 0 : 1) == 0 || t < 1) { if (tid < 256) needB[tid] = 0; }
      __syncthreads();
#endif
      if (t >= 1) {
         const float gT = thr[0];                         // threshold of the previous frame
         double myGen = LZERO, myWord = LZERO;
         if constexpr (NPT > 0) {
#pragma unroll
            for (int k = 0; k < NPT; k++) {
               const int hk = tid + k * NTHR;
               if (hk < nReg) {
                  const int4 ra = N.regRecA[hk];          // {node, kind | N << 4, offset of transP, fused node}: one coalesced load, nothing dependent behind it
                  const int4 rb = N.regRecB[hk];          // {score slots of states 2, 3, 4, nobody pulls the exit token}
                  const int n = ra.x;
                  const int4 ni = make_int4(ra.y, 0, ra.z, 0);
                  const int NS = (ni.x >> 4) & 255;
                  Tok s[DEC_MAXR];
                  const bool detached = rmax(k) < gT;     // DetachInst of the previous frame's pass 2
                  bool live = false;
                  s[0] = null_tok();
#pragma unroll
                  for (int i = 1; i < DEC_MAXR; i++) {
                     s[i] = null_tok();
                     if (i < NS && (i == 1 || !detached)) s[i] = (i == 1) ? xs[k * NTHR + tid] : rs[k][i - 2];
                  }
#pragma unroll
                  for (int i = 1; i < DEC_MAXR; i++) if (i < NS && s[i].like > LSMALL) live = true;
                  Tok exT = null_tok();
                  double mx = LZERO;
                  if (live) {
                     hmm_step1<DEC_MAXR, true>(a, ud, ni, tpBase + ni.z, s, gT, t, N.regRecF[hk].x, exT, mx, myWord, rb.x, rb.y, rb.z, rb.w >> 1);
                     if (mx > myGen) myGen = mx;
                     nLive++;
                  } else {
                     nDead++;
#pragma unroll
                     for (int i = 1; i < DEC_MAXR; i++) s[i] = null_tok();
                  }
#pragma unroll
                  for (int i = 2; i < DEC_MAXR; i++) rs[k][i - 2] = s[i];
                  xs[k * NTHR + tid] = exT;                // the exit token, until pass 2 has used it
                  if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four models at a time: all twelve interleaved, their temporaries spill)
                  if (!(rb.w & 1)) ex[n] = exT;            // (somebody pulls it from memory)
                  rmax(k) = (float)mx;                     // inst->max is a LogFloat (HRec.c:138); in memory only where -u wants to see it
                  if (a.maxActive > 0) imax[n] = (double)(float)mx;
               }
            }
         }
         if constexpr (HASG)
         for (int hk = nReg + tid; hk < N.nHmm; hk += NTHR) {
            const int n = N.hmmNodes[hk];
            const int4 ni = N.nodeInfo[n];
            const int NS = (ni.x >> 4) & 255, t0 = ni.y;
            Tok s[DEC_MAXN];
            const bool detached = imax[n] < gT;           // DetachInst of the previous frame's pass 2
            bool live = false;
            s[0] = null_tok();
#pragma unroll
            for (int i = 1; i < DEC_MAXN; i++) {
               s[i] = null_tok();
               // the entry token (i == 1) was pulled at the end of the previous frame's level phase
               if (i < NS && (i == 1 || !detached)) s[i] = tok[t0 + i - 1];
            }
#pragma unroll
            for (int i = 1; i < DEC_MAXN; i++) if (i < NS && s[i].like > LSMALL) live = true;
            Tok exT = null_tok();
            double mx = LZERO;
            if (live) {
               hmm_step1<DEC_MAXN>(a, ud, ni, tpBase + ni.z, s, gT, t, N.wdlk[n], exT, mx, myWord);
#pragma unroll
               for (int j = 1; j < DEC_MAXN; j++) if (j < NS) tok[t0 + j - 1] = s[j];
               if (mx > myGen) myGen = mx;
            } else if (detached) {
               for (int i = 1; i < NS; i++) tok[t0 + i - 1] = null_tok();
            }
            ex[n] = exT; imax[n] = (double)(float)mx;         // inst->max is a LogFloat (HRec.c:138)
         }
         DEC_STAMP(1);
         double genMax = myGen, wordMax = myWord;
         block_max2<NTHR>(genMax, wordMax, red, red2);
         if (tid == 0) {
            float w = (float)(wordMax - a.wordBeam); if (w < (float)LSMALL) w = (float)LSMALL;
            float g = (float)(genMax - a.genBeam); if (g < (float)LSMALL) g = (float)LSMALL;
            thr[0] = g; thr[1] = w;
         }
         __syncthreads();
      }
#ifdef DEC_NEED
      __syncthreads();
      if (t >= 1 && tid < 64 && (t % (DEC_NEED + 0 > 0 ? DEC_NEED + 0 : 1) == 0 || t == ud.T)) {      /* at the end of a block: its distinct states x its frames */
         unsigned int c = 0;
         for (int i = tid; i < 256; i += 64) c += __popc(needB[i]);
#pragma unroll
         for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
         const int blk = (DEC_NEED + 0 > 0 ? DEC_NEED + 0 : 1);
         needSum += (unsigned long long)c * (unsigned long long)(t % blk == 0 ? blk : t % blk);
      }
#endif
      DEC_STAMP(2);
      // ---- zero-time nodes, level by level (at t = 0: StartRecognition's propagation of the initial token)
      const float gT = thr[0], wT = thr[1];
      if constexpr (NPT > 0) {
         // word / null nodes with ONE predecessor, a register-resident model: stepped by its owner from the exit token in the register
         if (t >= 1) {
#pragma unroll
            for (int k = 0; k < NPT; k++) {
               const int hk = tid + k * NTHR;
               if (hk < nReg) {
                  const int4 fr = N.regFusedRec[hk];      // {fused node, its LDS slot (DecNet::zl), bits of its pronunciation probability, its column in the Path table or -1: a null node}
                  const int n = fr.x;
                  if (n >= 0) {
                     const Tok e0 = xs[k * NTHR + tid];
                     const float lm = N.regRecF[hk].y;
                     Tok st = null_tok();
                     if (e0.like > gT) {                   // pull_range over the one predecessor (a model: no word-end beam on its token)
                        const double c = e0.like + lm * a.lmScale;
                        if (c > gT) { st.like = c; st.lm = e0.lm + lm; st.path = e0.path; }
                     }
                     Tok e = null_tok();
                     if (st.like > LSMALL) {
                        e = st;
                        if (fr.w >= 0) {                      // StepWord2 (word_step2 with the node's constants from the record)
                           e.like += a.wordPen;
                           e.like += __int_as_float(fr.z) * a.prScale;
                           const size_t pid = (size_t)t * N.nWordNodes + fr.w;
                           // (written once, read by the final walk only: past the L2's retention)
                           __builtin_nontemporal_store(st.path, a.pathPrev + ud.path0 + pid); __builtin_nontemporal_store(e.like, a.pathLike + ud.path0 + pid);
                           __builtin_nontemporal_store(e.lm, a.pathLm + ud.path0 + pid);
                           e.path = (int)pid; e.lm = 0.0f;
                        }
                     }
                     if (a.maxActive > 0) imax[n] = (st.like > LSMALL) ? (double)(float)st.like : LZERO;      // (a word / null node's max is read by -u only)
                     put_ex(n, e, EXL ? fr.y : -1);
                  }
               }
               if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
         }
      }
      DEC_STAMP(3);
      for (int L = 0; L < N.nLevels; L++) {
         const int l0 = N.levelOff[L], lw = N.levelWide[L], l1 = N.levelOff[L + 1];
         for (int k = l0 + tid; k < lw; k += NTHR) {
            const int n = N.levelNodes[k];
            const int4 ni = N.nodeInfo[n];
            const int kind = ni.x & 15;
            int ak;
            Tok st = pull_range<EXL>(a, ex, N.predOff[n], N.predOff[n + 1], 1, gT, wT, &ak, &tie, wl, nullL, t * N.nWordNodes);
            if (t == 0 && n == N.initial) { st.like = 0.0; st.lm = 0.0f; st.path = -1; }
            Tok e = null_tok();
            if (kind == HTKAMD_NODE_HMM) {                 // tee model: StepHMM2
               const int NS = (ni.x >> 4) & 255;
               tok[ni.y] = st;
               e = ex[n];
               const double m2 = (st.like > imax[n]) ? (double)(float)st.like : imax[n];      // SetEntryState raises the instance's max
               if (t >= 1 && m2 < gT) { e = null_tok(); imax[n] = LZERO; }
               else {
                  imax[n] = m2;
                  if (st.like > LSMALL) {
                     const double c = st.like + tpBase[ni.z + (NS - 1)];
                     if (c > e.like) { e = st; e.like = c; }
                  }
               }
            } else if (!(st.like > LSMALL)) imax[n] = LZERO;
            else {
               imax[n] = (double)(float)st.like;
               e = (kind == HTKAMD_NODE_WORD) ? word_step2(a, ud, n, t, st) : st;
            }
            put_ex(n, e, (EXL && kind != HTKAMD_NODE_HMM) ? N.zl[n] : -1);
         }
         for (int k = lw; k < l1; k++) {                  // wide fan-in: the whole workgroup reduces one node
            const int n = N.levelNodes[k];
            int ak;
            Tok st = pull_range<EXL>(a, ex, N.predOff[n] + tid, N.predOff[n + 1], NTHR, gT, wT, &ak, &tie, wl, nullL, t * N.nWordNodes);
            // argmax over the workgroup: larger like, then smaller predecessor position
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
               const double ol = __shfl_xor(st.like, o); const int ok = __shfl_xor(ak, o);
               const float olm = __shfl_xor(st.lm, o); const int op = __shfl_xor(st.path, o);
               if (ol == st.like && ok != 0x7fffffff && ak != 0x7fffffff && (op != st.path || olm != st.lm)) tie = true;
               if (ol > st.like || (ol == st.like && ok < ak)) { st.like = ol; st.lm = olm; st.path = op; ak = ok; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { red[tid >> 6] = st.like; redk[tid >> 6] = ak; red2[tid >> 6] = __hiloint2double(__float_as_int(st.lm), st.path); }
            __syncthreads();
            if (tid == 0) {
               int bw = 0;
               for (int i = 1; i < NTHR / 64; i++) {
                  if (red[i] == red[bw] && redk[i] != 0x7fffffff && redk[bw] != 0x7fffffff && red2[i] != red2[bw]) tie = true;   // (lm, path) packed in red2
                  if (red[i] > red[bw] || (red[i] == red[bw] && redk[i] < redk[bw])) bw = i;
               }
               Tok b; b.like = red[bw]; b.lm = __int_as_float(__double2hiint(red2[bw])); b.path = __double2loint(red2[bw]);
               if (redk[bw] == 0x7fffffff) b = null_tok();
               if (t == 0 && n == N.initial) { b.like = 0.0; b.lm = 0.0f; b.path = -1; }
               Tok e = null_tok();
               imax[n] = (b.like > LSMALL) ? (double)(float)b.like : LZERO;
               if (b.like > LSMALL) e = (N.kind[n] == HTKAMD_NODE_WORD) ? word_step2(a, ud, n, t, b) : b;
               put_ex(n, e, EXL ? N.zl[n] : -1);
            }
         }
         __syncthreads();
      }
      DEC_STAMP(4);
      // ---- entry tokens of the emitting models for the next frame (SetEntryState from this frame's exits)
      if (t < T) {
         if constexpr (NPT > 0) {
            // four models in step, two predecessors of each per round: eight link words, then eight tokens in flight (see pull_range)
            constexpr int GM = (NPT % 4 == 0) ? 4 : 2;
#pragma unroll
            for (int k4 = 0; k4 < NPT; k4 += GM) {
               int2 pr[GM]; PullAcc pa[GM];
               int nmax = 0;
#pragma unroll
               for (int j = 0; j < GM; j++) {
                  const int hk = tid + (k4 + j) * NTHR;
                  pr[j] = (hk < nReg) ? N.regPred[hk] : make_int2(0, 0);
                  pa[j] = pull_start();
                  nmax = (pr[j].y - pr[j].x > nmax) ? pr[j].y - pr[j].x : nmax;
               }
               for (int i = 0; i < nmax; i += 2) {
                  int2 r[GM][2];
#pragma unroll
                  for (int j = 0; j < GM; j++)
#pragma unroll
                     for (int ii = 0; ii < 2; ii++) { const int kk = pr[j].x + i + ii; r[j][ii] = (EXL ? N.predRecRegL : N.predRecReg)[kk < pr[j].y ? kk : 0]; }
                  if constexpr (EXL) {
                     // the tokens come from LDS: taken and compared one by one (eight of them held at once spilled registers)
#pragma unroll
                     for (int j = 0; j < GM; j++)
#pragma unroll
                        for (int ii = 0; ii < 2; ii++) {
                           const int kk = pr[j].x + i + ii, x = r[j][ii].x;
                           if (kk >= pr[j].y) continue;
                           Tok e;
                           if (x & 0x40000000) {              // a WORD node: its likelihood from LDS, the rest of its token follows from the frame
                              const int wi = x & 0x1fffffff;
                              const double l = wl[wi];
                              e.like = l; e.lm = 0.0f; e.path = (l > LSMALL) ? t * N.nWordNodes + wi : -1;
                           } else if (x & 0x20000000) e = nullL[x & 0x1fffffff];
                           else e = ex[x & 0x1fffffff];
                           pull_fold(pa[j], a, e, x, __int_as_float(r[j][ii].y), kk, gT, wT);
                        }
                  } else {
                     Tok e[GM][2];
#pragma unroll
                     for (int j = 0; j < GM; j++)
#pragma unroll
                        for (int ii = 0; ii < 2; ii++) e[j][ii] = ex[r[j][ii].x & 0x1fffffff];
#pragma unroll
                     for (int j = 0; j < GM; j++)
#pragma unroll
                        for (int ii = 0; ii < 2; ii++) { const int kk = pr[j].x + i + ii; if (kk < pr[j].y) pull_fold(pa[j], a, e[j][ii], r[j][ii].x, __int_as_float(r[j][ii].y), kk, gT, wT); }
                  }
               }
#pragma unroll
               for (int j = 0; j < GM; j++) {
                  const int k = k4 + j, hk = tid + k * NTHR;
                  if (hk < nReg) {
                     if (pa[j].tb) tie = true;
                     const Tok en = pa[j].best;
                     xs[k * NTHR + tid] = en;
                     if (en.like > (double)rmax(k)) {
                        rmax(k) = (float)en.like;
                        if (a.maxActive > 0) imax[N.regRecA[hk].x] = (double)(float)en.like;
                     }
                  }
               }
               __builtin_amdgcn_sched_barrier(0);
            }
         }
         if constexpr (HASG)
         for (int hk = nReg + tid; hk < N.nHmm; hk += NTHR) {
            const int n = N.hmmNodes[hk];
            const int4 ni = N.nodeInfo[n];
            if ((ni.x >> 12) & 1) continue;                // tee models got theirs in the level phase
            int ak;
            const Tok en = pull_range<EXL>(a, ex, N.predOff[n], N.predOff[n + 1], 1, gT, wT, &ak, &tie, wl, nullL, t * N.nWordNodes);
            tok[ni.y] = en;
            if (en.like > imax[n]) imax[n] = (double)(float)en.like;      // SetEntryState: the entering token raises the instance's max
         }
         __syncthreads();
      }
      DEC_STAMP(5);
   }

#ifdef DEC_CLK
   if (tid == 0 && a.liveCnt) for (int i = 0; i < 6; i++) atomicAdd(a.liveCnt + 2 * a.nUtt + i, clk[i]);
#endif
#ifdef DEC_NEED
   if (tid == 0 && a.liveCnt) atomicAdd(a.liveCnt + 2 * a.nUtt + 6, needSum);
#endif
   if (tie) a.tieFlag[u] = 1;                  // (zeroed by the host before the launch)
   if (NPT > 0 && a.liveCnt) {
      unsigned long long l = nLive, dd = nDead;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o); dd += __shfl_xor(dd, o); }
      if ((tid & 63) == 0) { atomicAdd(a.liveCnt + 2 * u, l); atomicAdd(a.liveCnt + 2 * u + 1, dd); }
   }
   // ---- CompleteRecognition + LatFromPaths + TranscriptionFromLattice for the 1-best chain
   if (tid == 0) {
      Tok fin = ex[N.final];
      if constexpr (EXL) {
         const int zf = N.zl[N.final];
         if (zf >= 0) {
            if (zf & 0x40000000) { const double l = wl[zf & 0x3fffffff]; fin.like = l; fin.lm = 0.0f; fin.path = (l > LSMALL) ? T * N.nWordNodes + (zf & 0x3fffffff) : -1; }
            else fin = nullL[zf];
         }
      }
      const int fp = fin.path;
      int nW = 0;
      a.total[u] = LZERO; a.finalLm[u] = 0.0f;
      if (fp >= 0) {
         a.total[u] = fin.like; a.finalLm[u] = fin.lm;
         for (int p = fp; p >= 0; p = a.pathPrev[ud.path0 + p]) nW++;
         if (nW > a.maxWords) nW = -3;
         else {
            int w = nW;
            for (int p = fp; p >= 0;) {
               const int prev = a.pathPrev[ud.path0 + p];
               const int widx = p % N.nWordNodes, frame = p / N.nWordNodes;
               const double prlk = (prev >= 0) ? a.pathLike[ud.path0 + prev] : 0.0;
               const double wp = a.wordPen;
               const float plm = a.pathLm[ud.path0 + p];
               float aclike = (float)(a.pathLike[ud.path0 + p] - prlk - plm * a.lmScale - wp);
               const int node = N.wordNode[widx];
               const float pr = N.pronProb[node];
               aclike -= pr * a.prScale;
               const float sc = (float)((double)((aclike * 1.0f + plm * a.lmScale) + pr * a.prScale) + (double)a.wordPen);
               w--;
               a.wordPron[ud.out0 + w] = N.model[node];
               a.wordEnd[ud.out0 + w] = frame;
               a.wordStart[ud.out0 + w] = (prev >= 0) ? prev / N.nWordNodes : 0;
               a.wordScore[ud.out0 + w] = sc;
               a.wordLm[ud.out0 + w] = plm;
               a.wordAc[ud.out0 + w] = aclike;
               a.wordLike[ud.out0 + w] = a.pathLike[ud.path0 + p];
               p = prev;
            }
         }
      } else nW = -1;
      a.nWords[u] = nW;
   }
}

// state-major score block -> frame-major, per utterance: 32 x 32 tiles through LDS (both sides in whole lines)
__global__ __launch_bounds__(256) void k_score_transpose(const float *__restrict__ in, float *__restrict__ out, const DecUtt *utt, int ns)
{
   __shared__ float tile[32][33];
   const DecUtt ud = utt[blockIdx.z];
   const int T = ud.T, f0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
   if (f0 >= T) return;
   const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
   for (int r = ty; r < 32; r += 8) {
      const int sl = s0 + r, f = f0 + tx;
      tile[r][tx] = (sl < ns && f < T) ? in[ud.score0 + (size_t)sl * T + f] : 0.0f;
   }
   __syncthreads();
   for (int r = ty; r < 32; r += 8) {
      const int f = f0 + r, sl = s0 + tx;
      if (f < T && sl < ns) out[ud.score0 + (size_t)f * ns + sl] = tile[tx][r];
   }
}

int htkamd_launch_score_transpose(const float *in, float *out, const DecUtt *dUtt, int nUtt, int maxT, int ns, hipStream_t s)
{
   if (nUtt <= 0 || maxT <= 0 || ns <= 0) return HTKAMD_OK;
   hipLaunchKernelGGL(k_score_transpose, dim3((maxT + 31) / 32, (ns + 31) / 32, nUtt), dim3(256), 0, s, in, out, dUtt, ns);
   hipError_t e = hipGetLastError();
   if (e != hipSuccess) { htkamd_set_error("score_transpose: launch: %s", hipGetErrorString(e)); return HTKAMD_EHIP; }
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ host side
template <typename T> static int upv(htkamd_decoder *d, const std::vector<T> &v, const T **out)
{
   T *p = nullptr;
   HIPCHECK(hipMalloc((void **)&p, sizeof(T) * (v.size() ? v.size() : 1)));
   if (!v.empty()) HIPCHECK(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
   d->owned.push_back(p);
   *out = p;
   return HTKAMD_OK;
}

extern "C" int htkamd_decoder_set_order(htkamd_decoder *d, int mode)
{
   if (!d || mode < HTKAMD_ORDER_AUTO || mode > HTKAMD_ORDER_EXACT) { htkamd_set_error("decoder_set_order: bad argument"); return HTKAMD_EINVAL; }
   d->orderMode = mode;
   return HTKAMD_OK;
}
extern "C" int htkamd_decoder_last_tied(const htkamd_decoder *d) { return d ? d->lastTied : 0; }
extern "C" int htkamd_decoder_last_times(const htkamd_decoder *d, double *scoreMs, double *tokenMs)
{
   if (!d) { htkamd_set_error("decoder_last_times: NULL"); return HTKAMD_EINVAL; }
   if (scoreMs) *scoreMs = d->lastScoreMs;
   if (tokenMs) *tokenMs = d->lastTokenMs;
   return HTKAMD_OK;
}

extern "C" int htkamd_decoder_last_live(const htkamd_decoder *d, long long out[2])
{
   if (!d || !out) { htkamd_set_error("decoder_last_live: NULL"); return HTKAMD_EINVAL; }
   out[0] = d->lastLive[0]; out[1] = d->lastLive[1];
   return HTKAMD_OK;
}

extern "C" void htkamd_decoder_destroy(htkamd_decoder *d)
{
   if (!d) return;
   for (void *p : d->owned) (void)hipFree(p);
   for (hipEvent_t e : d->ev) if (e) (void)hipEventDestroy(e);
   for (void *p : d->ws) if (p) (void)hipFree(p);
   for (void *p : d->wsN) if (p) (void)hipFree(p);
   delete d;
}

// IsWd0Link (HNet.c:1663): the link leads to a word node, directly or through tee models
static bool is_wd0_link(const htkamd_net_desc *nd, const std::vector<unsigned char> &tee, int dst)
{
   if (nd->kind[dst] != HTKAMD_NODE_HMM) return true;
   if (!tee[dst]) return false;
   for (int k = nd->linkOff[dst]; k < nd->linkOff[dst + 1]; k++) if (is_wd0_link(nd, tee, nd->linkDest[k])) return true;
   return false;
}

static float like_to_word(const htkamd_net_desc *nd, const htkamd_model *m, const std::vector<unsigned char> &tee, int n, float scale)
{
   float best = (float)LZERO;
   for (int k = nd->linkOff[n]; k < nd->linkOff[n + 1]; k++) {
      const int dst = nd->linkDest[k];
      const bool zt = nd->kind[dst] != HTKAMD_NODE_HMM || tee[dst];
      if (!zt) continue;
      float like = nd->linkLike[k] * scale;
      if (like <= best) continue;
      if (nd->kind[dst] != HTKAMD_NODE_HMM) { if (like > best) best = like; }
      else {
         const int ti = m->h_hmmTrans[nd->model[dst]], NS = m->h_transN[ti];
         like += m->h_transP[m->h_transOff[ti] + (NS - 1)];
         like += like_to_word(nd, m, tee, dst, scale);
         if (like > best) best = like;
      }
   }
   return best;
}

extern "C" int htkamd_decoder_create(htkamd_model *m, const htkamd_net_desc *nd, float lmScale, htkamd_decoder **out)
{
   if (!m || !nd || !out) { htkamd_set_error("decoder_create: NULL argument"); return HTKAMD_EINVAL; }
   const int nN = nd->nNodes;
   std::vector<int> kind(nd->kind, nd->kind + nN), model(nd->model, nd->model + nN), tok0(nN), nodeN(nN, 2), nodeTp(nN, 0), nodeSt(nN, 0), wordIdx(nN, -1), hmmNodes;
   std::vector<float> pron(nd->pronProb, nd->pronProb + nN), wdlk(nN, (float)LZERO);
   std::vector<unsigned char> tee(nN, 0);
   std::vector<int> stateSlot(m->S, -1), wordNode;
   htkamd_decoder *d = new htkamd_decoder();
   d->m = m; d->d_usedStates = nullptr; d->hostModel = model;
   int nTok = 0, nW = 0;
   for (int n = 0; n < nN; n++) {
      tok0[n] = nTok;
      if (kind[n] == HTKAMD_NODE_HMM) {
         const int h = model[n];
         if (h < 0 || h >= m->H) { htkamd_set_error("decoder_create: node %d names model %d of %d", n, h, m->H); delete d; return HTKAMD_EINVAL; }
         const int ti = m->h_hmmTrans[h], NS = m->h_transN[ti];
         if (NS > DEC_MAXN) { htkamd_set_error("decoder_create: model with %d states (max %d)", NS, DEC_MAXN); delete d; return HTKAMD_EMODEL; }
         nodeN[n] = NS; nodeTp[n] = m->h_transOff[ti]; nodeSt[n] = m->h_hmmStateOff[h] / m->NSt;      // several streams: the table below lists a state's FIRST element only
         tee[n] = m->h_transP[m->h_transOff[ti] + (NS - 1)] > (float)LSMALL;
         nTok += NS - 1;
         hmmNodes.push_back(n);
         for (int j = 0; j < NS - 2; j++) {
            const int s = m->h_hmmState[m->h_hmmStateOff[h] + j * m->NSt];
            if (stateSlot[s] < 0) { stateSlot[s] = (int)d->usedStates.size(); d->usedStates.push_back(s); }
         }
      } else {
         nTok += 1;
         if (kind[n] == HTKAMD_NODE_WORD) { wordIdx[n] = nW++; wordNode.push_back(n); }
      }
   }
   // Zero-time sub-graph (word ends, null nodes, tee models) in topological order: Kahn's algorithm seeded in node order, successors
   // in link order.  This order is also the order of the zero-time senders of a frame (below).
   auto zt = [&](int n) { return kind[n] != HTKAMD_NODE_HMM || tee[n]; };
   std::vector<int> level(nN, -1), indeg(nN, 0), queue;
   for (int n = 0; n < nN; n++) if (zt(n)) for (int k = nd->linkOff[n]; k < nd->linkOff[n + 1]; k++) if (zt(nd->linkDest[k])) indeg[nd->linkDest[k]]++;
   for (int n = 0; n < nN; n++) if (zt(n) && indeg[n] == 0) { level[n] = 0; queue.push_back(n); }
   int nLevels = 0;
   for (size_t qi = 0; qi < queue.size(); qi++) {
      const int n = queue[qi];
      if (level[n] + 1 > nLevels) nLevels = level[n] + 1;
      for (int k = nd->linkOff[n]; k < nd->linkOff[n + 1]; k++) {
         const int dst = nd->linkDest[k];
         if (!zt(dst)) continue;
         if (level[n] + 1 > level[dst]) level[dst] = level[n] + 1;
         if (--indeg[dst] == 0) queue.push_back(dst);
      }
   }
   { int nz = 0; for (int n = 0; n < nN; n++) if (zt(n)) nz++;
     if ((int)queue.size() != nz) { htkamd_set_error("decoder_create: the network has a loop of word/null/tee nodes"); delete d; return HTKAMD_EMODEL; } }
   // Reverse CSR.  A node keeps the FIRST of several equally likely tokens (SetEntryState HRec.c:1303, strict >), so the order of a
   // node's predecessors is the order in which the senders are stepped within a frame: the emitting models' exit tokens go out first
   // (node order), then the zero-time nodes in their propagation order (HRec keeps its instance list in that order, ReOrderList
   // HRec.c:1152; oracle/orc_decode.c walks the same sequence); links of one sender stay in link order.
   std::vector<int> predOff(nN + 1, 0), predSrc(nd->nLinks);
   std::vector<float> predLike(nd->nLinks);
   for (int k = 0; k < nd->nLinks; k++) predOff[nd->linkDest[k] + 1]++;
   for (int n = 0; n < nN; n++) predOff[n + 1] += predOff[n];
   {
      std::vector<int> fill(predOff.begin(), predOff.end() - 1);
      auto send = [&](int n) {
         for (int k = nd->linkOff[n]; k < nd->linkOff[n + 1]; k++) {
            const int at = fill[nd->linkDest[k]]++;
            predSrc[at] = n | (kind[n] != HTKAMD_NODE_HMM ? (int)0x80000000 : 0);      // bit 31: word/null predecessor (word-end beam applies)
            predLike[at] = nd->linkLike[k];
         }
      };
      for (int n = 0; n < nN; n++) if (!zt(n)) send(n);
      for (int n : queue) send(n);
   }
   // Register-resident models (k_decode<NPT > 0>): plain models of at most three emitting states come first in hmmNodes, up to what
   // DEC_REG_MAXNPT per thread hold; a word / null node of level 0 with ONE predecessor, such a model, is stepped by that model's owner.
   std::vector<int> regFused, fusedOf(nN, -1);
   std::vector<int4> regRecA, regRecB;
   std::vector<float2> regRecF;
   std::vector<float> regFusedLike;
   std::vector<unsigned char> regNoEx, isFused(nN, 0);
   int nReg = 0;
   if (!getenv("HTKAMD_DECODE_NOREG")) {
      std::vector<int> reg, rest;
      for (int n : hmmNodes) ((!tee[n] && nodeN[n] <= 5 && (int)reg.size() < DEC_REG_MAXNPT * DEC_REG_THREADS) ? reg : rest).push_back(n);
      nReg = (int)reg.size();
      // most predecessors first: the entry pulls of a wavefront run until its longest list is done (two predecessors a round), so lists of one
      // length belong together -- thread t's k-th model is reg[t + k NTHR], the 1 024 longest lists are everybody's first model, and so on
      // (6 000-word bigram network: 20 rounds a frame in storage order, 11 sorted)
      std::stable_sort(reg.begin(), reg.end(), [&](int x, int y) { return predOff[x + 1] - predOff[x] > predOff[y + 1] - predOff[y]; });
      hmmNodes = reg; hmmNodes.insert(hmmNodes.end(), rest.begin(), rest.end());
      std::vector<int> slotOf(nN, -1);
      for (int k = 0; k < nReg; k++) slotOf[reg[k]] = k;
      regFused.assign(nReg, -1); regFusedLike.assign(nReg, 0.0f); regNoEx.assign(nReg, 0);
      for (int w = 0; w < nN; w++) {
         if (kind[w] == HTKAMD_NODE_HMM || w == nd->initial || level[w] != 0 || predOff[w + 1] - predOff[w] != 1) continue;
         const int pnode = predSrc[predOff[w]] & 0x7fffffff;
         const int k = slotOf[pnode];
         if (k < 0 || regFused[k] >= 0) continue;
         regFused[k] = w; regFusedLike[k] = predLike[predOff[w]]; isFused[w] = 1;
      }
      for (int k = 0; k < nReg; k++) {
         const int n = reg[k];
         regNoEx[k] = regFused[k] >= 0 && nd->linkOff[n + 1] - nd->linkOff[n] == 1;
      }
      regRecA.resize(nReg); regRecB.resize(nReg); regRecF.resize(nReg);
   }
   std::vector<int> levelOff(nLevels + 1, 0), levelWide(nLevels, 0), levelNodes;
   for (int L = 0; L < nLevels; L++) {
      levelOff[L] = (int)levelNodes.size();
      for (int n = 0; n < nN; n++) if (zt(n) && level[n] == L && predOff[n + 1] - predOff[n] < DEC_WIDE && !isFused[n]) levelNodes.push_back(n);
      levelWide[L] = (int)levelNodes.size();
      for (int n = 0; n < nN; n++) if (zt(n) && level[n] == L && predOff[n + 1] - predOff[n] >= DEC_WIDE) {
         if (kind[n] == HTKAMD_NODE_HMM) { htkamd_set_error("decoder_create: tee model with %d predecessors", predOff[n + 1] - predOff[n]); delete d; return HTKAMD_EMODEL; }
         levelNodes.push_back(n);
      }
   }
   levelOff[nLevels] = (int)levelNodes.size();
   if (getenv("HTKAMD_DECODE_VERBOSE"))
      for (int L = 0; L < nLevels; L++) fprintf(stderr, "decoder_create: level %d: %d narrow + %d wide zero-time nodes (register models %d, fused nodes %d)\n", L, levelWide[L] - levelOff[L], levelOff[L + 1] - levelWide[L], nReg, (int)std::count(isFused.begin(), isFused.end(), 1));
   // ... and the same lists with every zero-time node in them, for the kernels that keep no tokens in registers (k_decode_n)
   std::vector<int> levelOffA(nLevels + 1, 0), levelWideA(nLevels, 0), levelNodesA;
   for (int L = 0; L < nLevels; L++) {
      levelOffA[L] = (int)levelNodesA.size();
      for (int n = 0; n < nN; n++) if (zt(n) && level[n] == L && predOff[n + 1] - predOff[n] < DEC_WIDE) levelNodesA.push_back(n);
      levelWideA[L] = (int)levelNodesA.size();
      for (int n = 0; n < nN; n++) if (zt(n) && level[n] == L && predOff[n + 1] - predOff[n] >= DEC_WIDE) levelNodesA.push_back(n);
   }
   levelOffA[nLevels] = (int)levelNodesA.size();
   for (int n = 0; n < nN; n++) {
      bool wd0 = false;
      if (kind[n] == HTKAMD_NODE_HMM) for (int k = nd->linkOff[n]; k < nd->linkOff[n + 1]; k++) if (is_wd0_link(nd, tee, nd->linkDest[k])) wd0 = true;   // n_wd0 (HNet.c:3626-3631)
      if (wd0) wdlk[n] = like_to_word(nd, m, tee, n, lmScale);
   }
   for (int k = 0; k < nReg; k++) {
      const int n = hmmNodes[k], h = model[n];
      int sl[3] = {0, 0, 0};
      for (int j = 0; j < nodeN[n] - 2 && j < 3; j++) sl[j] = stateSlot[m->h_hmmState[m->h_hmmStateOff[h] + j * m->NSt]];
      regRecA[k] = make_int4(n, kind[n] | (nodeN[n] << 4), nodeTp[n], regFused[k]);
      // CreateSEIndex's ranges of the model's states 2 .. 4 and of its exit state, (lo, hi) in three bits each (hmm_step1<.., SLOTS>)
      int se = 0;
      {
         const int NS = nodeN[n];
         const float *tp = m->h_transP + nodeTp[n];
         for (int j = 2; j <= 4; j++) {
            int lo = 1, hi = NS - 1;
            if (j < NS) {
               while (lo < NS && !(tp[(lo - 1) * NS + (j - 1)] > LSMALL)) lo++;
               while (hi > 1 && !(tp[(hi - 1) * NS + (j - 1)] > LSMALL)) hi--;
               if (lo > hi) { lo = 1; hi = NS - 1; }
            }
            se |= (lo & 7) << (6 * (j - 2)) | (hi & 7) << (6 * (j - 2) + 3);
         }
         int lo = 2, hi = NS - 1;
         while (lo < NS && !(tp[(lo - 1) * NS + (NS - 1)] > LSMALL)) lo++;
         while (hi > 1 && !(tp[(hi - 1) * NS + (NS - 1)] > LSMALL)) hi--;
         if (lo > hi) { lo = 2; hi = NS - 1; }
         se |= (lo & 7) << 18 | (hi & 7) << 21;
      }
      regRecB[k] = make_int4(sl[0], sl[1], sl[2], (int)regNoEx[k] | (se << 1));
      regRecF[k] = make_float2(wdlk[n], regFusedLike[k]);
   }
   std::vector<int4> nodeInfo(nN);
   for (int n = 0; n < nN; n++) nodeInfo[n] = make_int4(kind[n] | (nodeN[n] << 4) | ((int)tee[n] << 12), tok0[n], nodeTp[n], nodeSt[n]);
   DecNet &N = d->net;
   memset(&N, 0, sizeof(N));
   int rc0 = HTKAMD_OK;
   N.nReg = nReg;
   std::vector<int2> regPred(nReg), predRec(predSrc.size() ? predSrc.size() : 1, make_int2(0, 0));
   for (size_t k = 0; k < predSrc.size(); k++) { int b; memcpy(&b, &predLike[k], 4); predRec[k] = make_int2(predSrc[k], b); }
   // LDS slots of the zero-time nodes the register-resident models pull from (k_decode<.., EXL>)
   std::vector<int> zl(nN, -1), regFusedZl(nReg > 0 ? nReg : 1, -1);
   std::vector<int2> predRecL(predRec);
   int nNullLds = 0;
   for (int k = 0; k < nReg; k++)
      for (int q = predOff[hmmNodes[k]]; q < predOff[hmmNodes[k] + 1]; q++) {
         const int src = predSrc[q] & 0x7fffffff;
         if (kind[src] == HTKAMD_NODE_WORD) zl[src] = 0x40000000 | wordIdx[src];
         else if (kind[src] != HTKAMD_NODE_HMM && zl[src] < 0 && nNullLds < DEC_NULL_LDS) zl[src] = nNullLds++;
      }
   for (size_t q = 0; q < predSrc.size(); q++) {
      const int src = predSrc[q] & 0x7fffffff;
      if (zl[src] >= 0) predRecL[q].x = (int)0x80000000 | ((zl[src] & 0x40000000) ? zl[src] : (0x20000000 | zl[src]));
   }
   // the register-resident models' lists once more, in the models' order (neighbouring lanes read neighbouring records), both forms: regPred indexes these
   std::vector<int2> predRecReg, predRecRegL;
   for (int k = 0; k < nReg; k++) {
      const int n = hmmNodes[k];
      regPred[k] = make_int2((int)predRecReg.size(), (int)predRecReg.size() + predOff[n + 1] - predOff[n]);
      for (int q = predOff[n]; q < predOff[n + 1]; q++) { predRecReg.push_back(predRec[q]); predRecRegL.push_back(predRecL[q]); }
   }
   if (predRecReg.empty()) { predRecReg.push_back(make_int2(0, 0)); predRecRegL.push_back(make_int2(0, 0)); }
   std::vector<int4> regFusedRec(nReg > 0 ? nReg : 1, make_int4(-1, -1, 0, -1));
   for (int k = 0; k < nReg; k++)
      if (regFused[k] >= 0) {
         const int w = regFused[k];
         int pb; memcpy(&pb, &pron[w], 4);
         regFusedZl[k] = zl[w];
         regFusedRec[k] = make_int4(w, zl[w], pb, kind[w] == HTKAMD_NODE_WORD ? wordIdx[w] : -1);
      }
   N.nNullLds = nNullLds;
   if ((rc0 = upv(d, predRec, &N.predRec)) || (rc0 = upv(d, predRecL, &N.predRecL)) || (rc0 = upv(d, zl, &N.zl)) || (rc0 = upv(d, regFusedZl, &N.regFusedZl)) || (rc0 = upv(d, regFusedRec, &N.regFusedRec)) || (rc0 = upv(d, predRecReg, &N.predRecReg)) || (rc0 = upv(d, predRecRegL, &N.predRecRegL)) ||
       (nReg > 0 && (rc0 = upv(d, regPred, &N.regPred)))) { htkamd_decoder_destroy(d); return rc0; }
   if (nReg > 0 && ((rc0 = upv(d, regFused, &N.regFused)) || (rc0 = upv(d, regFusedLike, &N.regFusedLike)) || (rc0 = upv(d, regNoEx, &N.regNoEx)) || (rc0 = upv(d, regRecA, &N.regRecA)) || (rc0 = upv(d, regRecB, &N.regRecB)) || (rc0 = upv(d, regRecF, &N.regRecF)))) { htkamd_decoder_destroy(d); return rc0; }
   N.nNodes = nN; N.nHmm = (int)hmmNodes.size(); N.nLevels = nLevels; N.nWordNodes = nW > 0 ? nW : 1; N.initial = nd->initial; N.final = nd->final; N.nTok = nTok; N.nTpFloats = m->h_transOff[m->nT];
   int rc;
   if ((rc = upv(d, kind, &N.kind)) || (rc = upv(d, model, &N.model)) || (rc = upv(d, pron, &N.pronProb)) || (rc = upv(d, predOff, &N.predOff)) ||
       (rc = upv(d, predSrc, &N.predSrc)) || (rc = upv(d, predLike, &N.predLike)) || (rc = upv(d, tok0, &N.tok0)) || (rc = upv(d, hmmNodes, &N.hmmNodes)) ||
       (rc = upv(d, nodeN, &N.nodeN)) || (rc = upv(d, nodeTp, &N.nodeTp)) || (rc = upv(d, nodeSt, &N.nodeSt)) || (rc = upv(d, tee, &N.nodeTee)) ||
       (rc = upv(d, wdlk, &N.wdlk)) || (rc = upv(d, wordIdx, &N.wordIdx)) || (rc = upv(d, levelOff, &N.levelOff)) || (rc = upv(d, levelNodes, &N.levelNodes)) ||
       (rc = upv(d, levelWide, &N.levelWide)) || (rc = upv(d, levelOffA, &N.levelOffAll)) || (rc = upv(d, levelNodesA, &N.levelNodesAll)) || (rc = upv(d, levelWideA, &N.levelWideAll)) || (rc = upv(d, stateSlot, &N.stateSlot)) || (rc = upv(d, wordNode, &N.wordNode)) ||
       (rc = upv(d, nodeInfo, &N.nodeInfo))) { htkamd_decoder_destroy(d); return rc; }
   {
      std::vector<int> hs((size_t)(m->h_hmmStateOff[m->H] / m->NSt));
      for (size_t i = 0; i < hs.size(); i++) hs[i] = m->h_hmmState[i * m->NSt];      // a state's first (state, stream) element: the scorer sums its streams (ScoreArgs::NSt)
      if ((rc = upv(d, hs, &N.hmmState))) { htkamd_decoder_destroy(d); return rc; }
      const int *us = nullptr;
      if ((rc = upv(d, d->usedStates, &us))) { htkamd_decoder_destroy(d); return rc; }
      d->d_usedStates = (int *)us;
   }
   {
      // the network's own link order for the exact-order kernels (decode_ord.hip)
      std::vector<int> lo(nd->linkOff, nd->linkOff + nN + 1), ld(nd->linkDest, nd->linkDest + nd->nLinks), nTr0(nN, 0);
      std::vector<float> ll(nd->linkLike, nd->linkLike + nd->nLinks);
      std::vector<unsigned char> dup(nN, 0);
      for (int n = 0; n < nN; n++) {
         int c = 0;
         for (int k = lo[n]; k < lo[n + 1] && zt(ld[k]); k++) c++;
         nTr0[n] = c;
         std::vector<int> ds(ld.begin() + lo[n], ld.begin() + lo[n + 1]);
         std::sort(ds.begin(), ds.end());
         dup[n] = std::adjacent_find(ds.begin(), ds.end()) != ds.end();
      }
      if ((rc = upv(d, lo, &N.linkOff)) || (rc = upv(d, ld, &N.linkDest)) || (rc = upv(d, ll, &N.linkLike)) || (rc = upv(d, nTr0, &N.nTr0)) ||
          (rc = upv(d, dup, &N.dupDest))) { htkamd_decoder_destroy(d); return rc; }
   }
   N.transP = m->d_transP;
   *out = d;
   return HTKAMD_OK;
}

extern "C" int htkamd_decoder_run(htkamd_decoder *d, const htkamd_decode_config *cfg, const float *dX, const int *frameOff, int nUtt,
                                  int maxWords, int *nWords, int *wordPron, int *wordStart, int *wordEnd, float *wordScore, float *wordLm,
                                  double *total, void *stream)
{
   htkamd_decode_out o;
   memset(&o, 0, sizeof(o));
   o.nWords = nWords; o.wordPron = wordPron; o.wordStart = wordStart; o.wordEnd = wordEnd; o.wordScore = wordScore; o.wordLm = wordLm; o.total = total;
   return htkamd_decoder_run_out(d, cfg, dX, frameOff, nUtt, maxWords, &o, stream);
}

extern "C" int htkamd_decoder_run_out(htkamd_decoder *d, const htkamd_decode_config *cfg, const float *dX, const int *frameOff, int nUtt,
                                      int maxWords, const htkamd_decode_out *out, void *stream)
{
   if (!out) { htkamd_set_error("decoder_run: bad argument"); return HTKAMD_EINVAL; }
   int *nWords = out->nWords, *wordPron = out->wordPron, *wordStart = out->wordStart, *wordEnd = out->wordEnd;
   float *wordScore = out->wordScore, *wordLm = out->wordLm, *wordAc = out->wordAc;
   double *total = out->total, *wordLike = out->wordLike;
   float *finalLm = out->finalLm;
   if (!d || !cfg || !frameOff || nUtt < 0 || maxWords < 1 || !nWords || !wordPron || !wordStart || !wordEnd || !wordScore || !total) {
      htkamd_set_error("decoder_run: bad argument"); return HTKAMD_EINVAL;
   }
   if (nUtt == 0) return HTKAMD_OK;
   hipStream_t s = (hipStream_t)stream;
   htkamd_model *m = d->m;
   const DecNet &N = d->net;
   const int ns = (int)d->usedStates.size();
   // (the matrix-core kernels build their frames' operand once per task: four times as many states per task for them, as in forward-backward)
   const int FR = SCORE_TILE_FRAMES, SL = (cfg->scoreMode & (HTKAMD_SCORE_BF16 | HTKAMD_SCORE_MFMA)) ? SCORE_TASK_SLOTS_WIDE : SCORE_TASK_SLOTS;
   if (!d->ev[0]) for (int i = 0; i < 4; i++) HIPCHECK(hipEventCreate(&d->ev[i]));
   d->lastScoreMs = d->lastTokenMs = 0.0f;
   d->lastLive[0] = d->lastLive[1] = 0;
   int orderMode = d->orderMode;
   if (const char *ev = getenv("HTKAMD_DECODE_ORDER")) orderMode = !strcmp(ev, "fast") ? HTKAMD_ORDER_FAST : !strcmp(ev, "exact") ? HTKAMD_ORDER_EXACT : HTKAMD_ORDER_AUTO;
   d->lastTied = 0;
   // chunk the batch so that the per-utterance work space (scores, tokens, path table) stays under ~24 GB
   int u0 = 0;
   while (u0 < nUtt) {
      size_t bytes = 0; int u1 = u0;
      while (u1 < nUtt) {
         const size_t T = (size_t)(frameOff[u1 + 1] - frameOff[u1]);
         const size_t b = (size_t)ns * T * 8 + (size_t)N.nTok * 16 + (size_t)N.nNodes * 24 + (T + 1) * (size_t)N.nWordNodes * 16;
         if (u1 > u0 && bytes + b > ((size_t)24 << 30)) break;
         bytes += b; u1++;
      }
      const int nu = u1 - u0;
      std::vector<DecUtt> utt(nu);
      std::vector<ScoreTask> tasks;
      size_t score = 0, tok = 0, node = 0, path = 0;
      for (int k = 0; k < nu; k++) {
         DecUtt &ud = utt[k];
         ud.T = frameOff[u0 + k + 1] - frameOff[u0 + k]; ud.frame0 = frameOff[u0 + k]; ud.status = HTKAMD_UTT_OK; ud.idx = k;
         ud.score0 = score; ud.tok0 = tok; ud.node0 = node; ud.path0 = path; ud.out0 = (size_t)k * maxWords;
         for (int ti = 0; ti * FR < ud.T; ti++)
            for (int ch = 0; ch * SL < ns; ch++) {
               ScoreTask tk;
               tk.frame0 = ud.frame0 + ti * FR; tk.nFrames = std::min(FR, ud.T - ti * FR);
               tk.slot0 = ch * SL; tk.nSlots = std::min(SL, ns - ch * SL); tk.outSlot0 = ch * SL; tk.ldo = ud.T;
               tk.outBase = ud.score0 + (size_t)ti * FR;
               tasks.push_back(tk);
            }
         score += (size_t)ns * ud.T; tok += (size_t)N.nTok; node += (size_t)N.nNodes; path += (size_t)(ud.T + 1) * N.nWordNodes;
      }
      void *dScore = nullptr, *dTok = nullptr, *dEx = nullptr, *dImax = nullptr;
      void *dPPrev = nullptr, *dPLike = nullptr, *dPLm = nullptr, *dUtt = nullptr, *dTasks = nullptr, *dOutI = nullptr, *dOutF = nullptr, *dTot = nullptr, *dOutD = nullptr;
      int rc = HTKAMD_OK;
      int wsi = 0;
      auto A = [&](void **p, size_t n) {                 // the decoder's own buffers, grown (not shrunk) as batches ask
         const int i = wsi++;
         if (rc) return;
         if (n < 1) n = 1;
         if (d->wsCap[i] < n) {
            if (d->ws[i]) { (void)hipStreamSynchronize(s); (void)hipFree(d->ws[i]); d->ws[i] = nullptr; d->wsCap[i] = 0; }
            const size_t want = n + n / 8;
            hipError_t e = hipMalloc(&d->ws[i], want);
            if (e != hipSuccess) { htkamd_set_error("decoder_run: hipMalloc(%zu): %s", want, hipGetErrorString(e)); rc = HTKAMD_ENOMEM; d->ws[i] = nullptr; return; }
            d->wsCap[i] = want;
         }
         *p = d->ws[i];
      };
      void *dScoreT = nullptr;
      A(&dScore, score * 4); A(&dScoreT, score * 4); A(&dTok, tok * sizeof(Tok)); A(&dEx, node * sizeof(Tok)); A(&dImax, node * 8);
      A(&dPPrev, path * 4); A(&dPLike, path * 8); A(&dPLm, path * 4);
      A(&dUtt, sizeof(DecUtt) * nu); A(&dTasks, sizeof(ScoreTask) * tasks.size() + sizeof(int));
      A(&dOutI, sizeof(int) * ((size_t)nu * maxWords * 3 + nu)); A(&dOutF, sizeof(float) * ((size_t)nu * maxWords * 3 + nu)); A(&dTot, sizeof(double) * nu);
      A(&dOutD, sizeof(double) * (size_t)nu * maxWords);
      void *dTie = nullptr;
      const size_t tieBytes = (sizeof(int) * nu + 7) & ~(size_t)7;
      A(&dTie, tieBytes + sizeof(unsigned long long) * (2 * (size_t)nu + 8));
      std::vector<int> hI; std::vector<float> hF; std::vector<double> hT, hD;
      if (!rc) {
         hipError_t e;
         if ((e = hipMemcpyAsync(dUtt, utt.data(), sizeof(DecUtt) * nu, hipMemcpyHostToDevice, s)) != hipSuccess ||
             (e = hipMemcpyAsync(dTasks, tasks.data(), sizeof(ScoreTask) * tasks.size(), hipMemcpyHostToDevice, s)) != hipSuccess) {
            htkamd_set_error("decoder_run: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP;
         }
      }
      if (!rc) (void)hipEventRecord(d->ev[0], s);
      if (!rc) {
         ScoreArgs sa;
         sa.tasks = (const ScoreTask *)dTasks; sa.nTasks = (int)tasks.size(); sa.X = dX; sa.slotState = d->d_usedStates; sa.out = (float *)dScore;
         sa.stateCompOff = m->d_stateCompOff; sa.compGauss = m->d_compGauss; sa.compLogWt = m->d_compLogWt;
         sa.gparam = m->d_gparam; sa.PS = m->PS; sa.D = m->D; sa.minLogExp = m->minLogExp;
         sa.laddTab = m->d_laddTab; sa.taskCounter = (int *)((char *)dTasks + sizeof(ScoreTask) * tasks.size());
         sa.mfmaTab = m->d_mfmaTab; sa.stateTileOff = m->d_stateTileOff; sa.bf16Tab = m->d_bf16Tab; sa.var = m->d_var;
         sa.NSt = m->NSt; sa.streamWt = m->d_streamWt;
         if (m->NSt > 1 && cfg->scoreMode != HTKAMD_SCORE_EXACT) { htkamd_set_error("decoder_run: multi-stream sets are scored in the exact mode only"); rc = HTKAMD_EINVAL; }
         else if (m->tiedMix) rc = htkamd_tm_score_block(m, sa, frameOff[u1], m->tmBeam, s);
         else if (cfg->scoreMode != HTKAMD_SCORE_EXACT && cfg->scoreMode != HTKAMD_SCORE_MFMA && cfg->scoreMode != HTKAMD_SCORE_BF16) { htkamd_set_error("decoder_run: unknown score mode %d", cfg->scoreMode); rc = HTKAMD_EINVAL; }
         else rc = htkamd_launch_score(cfg->scoreMode, m, sa, s);   // exact: the decoded path is the reference's; matrix-core modes: tolerance class
      }
      if (!rc) {
         int maxT = 0;
         for (int k = 0; k < nu; k++) maxT = std::max(maxT, utt[k].T);
         rc = htkamd_launch_score_transpose((const float *)dScore, (float *)dScoreT, (const DecUtt *)dUtt, nu, maxT, ns, s);
      }
      if (!rc) { (void)hipEventRecord(d->ev[1], s); (void)hipEventRecord(d->ev[2], s); }
      DecArgs a;
      memset(&a, 0, sizeof(a));
      if (!rc) {
         a.net = N; a.utt = (const DecUtt *)dUtt; a.nUtt = nu; a.score = (const float *)dScoreT; a.ns = ns;
         a.tok = (Tok *)dTok; a.ex = (Tok *)dEx; a.imax = (double *)dImax;
         a.pathPrev = (int *)dPPrev; a.pathLike = (double *)dPLike; a.pathLm = (float *)dPLm;
         a.genBeam = cfg->genBeam; a.wordBeam = cfg->wordBeam; a.lmScale = cfg->lmScale; a.wordPen = cfg->wordPen; a.prScale = cfg->prScale; a.maxActive = cfg->maxActive;
         a.maxWords = maxWords;
         int *oi = (int *)dOutI;
         a.nWords = oi; a.wordPron = oi + nu; a.wordStart = a.wordPron + (size_t)nu * maxWords; a.wordEnd = a.wordStart + (size_t)nu * maxWords;
         a.wordScore = (float *)dOutF; a.wordLm = (float *)dOutF + (size_t)nu * maxWords; a.wordAc = (float *)dOutF + (size_t)nu * maxWords * 2; a.finalLm = (float *)dOutF + (size_t)nu * maxWords * 3; a.total = (double *)dTot;
         a.wordLike = (double *)dOutD;
         a.tieFlag = (int *)dTie;
         a.liveCnt = (unsigned long long *)((char *)dTie + tieBytes);
         (void)hipMemsetAsync(dTie, 0, tieBytes + sizeof(unsigned long long) * (2 * (size_t)nu + 8), s);
         // tokens of the plain models in registers where the network has such models (DecNet::nReg), NPT of them per thread
         const int npt = (N.nReg + DEC_REG_THREADS - 1) / DEC_REG_THREADS;
         const bool hasG = N.nReg < N.nHmm;
         // ... and the exit tokens they pull in LDS where xs, the word ends' likelihoods and the transition matrices fit beside the static 3 KB
#define DEC_LAUNCH_REG(NPT_) do { \
            const size_t ldsX_ = sizeof(Tok) * (size_t)(NPT_) * DEC_REG_THREADS + sizeof(double) * (size_t)N.nWordNodes + sizeof(float) * (size_t)N.nTpFloats; \
            const bool exl_ = ldsX_ + 3072 <= 160 * 1024 && !getenv("HTKAMD_DECODE_NOEXL"); \
            const size_t lds_ = exl_ ? ldsX_ : (sizeof(Tok) + sizeof(float)) * (size_t)(NPT_) * DEC_REG_THREADS; \
            const void *fn_ = exl_ ? (hasG ? (const void *)k_decode<NPT_, DEC_REG_THREADS, true, true> : (const void *)k_decode<NPT_, DEC_REG_THREADS, false, true>) \
                                   : (hasG ? (const void *)k_decode<NPT_, DEC_REG_THREADS, true, false> : (const void *)k_decode<NPT_, DEC_REG_THREADS, false, false>); \
            (void)hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
            void *args_[] = {(void *)&a}; \
            (void)hipLaunchKernel(fn_, dim3(nu), dim3(DEC_REG_THREADS), args_, lds_, s); } while (0)
         if (npt == 0) hipLaunchKernelGGL((k_decode<0, DEC_THREADS, true>), dim3(nu), dim3(DEC_THREADS), 0, s, a);
         else if (npt <= 2) DEC_LAUNCH_REG(2);
         else if (npt <= 4) DEC_LAUNCH_REG(4);
         else DEC_LAUNCH_REG(DEC_REG_MAXNPT);
#undef DEC_LAUNCH_REG
         hipError_t e = hipGetLastError();
         if (e != hipSuccess) { htkamd_set_error("decoder_run: launch: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP; }
         (void)hipEventRecord(d->ev[3], s);
      }
      if (!rc && orderMode != HTKAMD_ORDER_FAST) {
         // the utterances in which two equally likely tokens with different histories met (or all of them: HTKAMD_ORDER_EXACT) once more,
         // in the order of HRec's instance list (decode_ord.hip); their results replace k_decode's
         std::vector<int> flags(nu, 1);
         if (orderMode == HTKAMD_ORDER_AUTO) {
            hipError_t e;
            if ((e = hipMemcpyAsync(flags.data(), dTie, sizeof(int) * nu, hipMemcpyDeviceToHost, s)) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) {
               htkamd_set_error("decoder_run: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP;
            }
         }
         std::vector<DecUtt> sel;
         size_t opath = 0;
         const int pathExtra = 64;
         for (int k = 0; k < nu && !rc; k++)
            if (flags[k]) {
               DecUtt ud = utt[k];
               ud.path0 = opath;
               opath += 3 * ((size_t)(ud.T + 1) * N.nWordNodes) + pathExtra;      // StepWord2 "may be repeated" (HRec.c:1046): room for every word node thrice per frame
               sel.push_back(ud);
            }
         if (!rc && !sel.empty()) {
            const int nSel = (int)sel.size();
            const int seqCap = 8 * N.nNodes + 1024;          // appends of one frame's pass 2 (attaches + moves) on top of the live instances
            void *dSel = nullptr, *dSeq = nullptr, *dPos = nullptr, *dOoo = nullptr, *oPrev = nullptr, *oLike = nullptr, *oLm = nullptr, *oNode = nullptr, *oFrame = nullptr;
            A(&dSel, sizeof(DecUtt) * nSel); A(&dSeq, sizeof(int) * (size_t)nSel * 2 * seqCap); A(&dPos, sizeof(int) * node); A(&dOoo, node);
            A(&oPrev, opath * 4); A(&oLike, opath * 8); A(&oLm, opath * 4); A(&oNode, opath * 4); A(&oFrame, opath * 4);
            if (!rc) {
               hipError_t e = hipMemcpyAsync(dSel, sel.data(), sizeof(DecUtt) * nSel, hipMemcpyHostToDevice, s);
               if (e != hipSuccess) { htkamd_set_error("decoder_run: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP; }
            }
            if (!rc) {
               OrdArgs oa;
               oa.d = a; oa.d.utt = (const DecUtt *)dSel; oa.d.nUtt = nSel;
               oa.d.pathPrev = (int *)oPrev; oa.d.pathLike = (double *)oLike; oa.d.pathLm = (float *)oLm;
               oa.seq = (int *)dSeq; oa.seqCap = seqCap; oa.pos = (int *)dPos; oa.ooo = (unsigned char *)dOoo;
               oa.pathNode = (int *)oNode; oa.pathFrame = (int *)oFrame; oa.pathExtra = pathExtra;
               oa.keepFast = orderMode == HTKAMD_ORDER_AUTO ? 1 : 0;
               // (the path capacity the kernel assumes per utterance is 3 (T + 1) nWordNodes + pathExtra: see opath above)
               rc = htkamd_launch_decode_ord(oa, nSel, s);
            }
            d->lastTied += nSel;
         }
      }
      std::vector<unsigned long long> hLive(2 * (size_t)nu + 8, 0);
      if (!rc) (void)hipMemcpyAsync(hLive.data(), (char *)dTie + tieBytes, sizeof(unsigned long long) * hLive.size(), hipMemcpyDeviceToHost, s);
      if (!rc) {
         hI.resize((size_t)nu * maxWords * 3 + nu); hF.resize((size_t)nu * maxWords * 3 + nu); hT.resize(nu); hD.resize((size_t)nu * maxWords);
         hipError_t e;
         if ((e = hipMemcpyAsync(hI.data(), dOutI, sizeof(int) * hI.size(), hipMemcpyDeviceToHost, s)) != hipSuccess ||
             (e = hipMemcpyAsync(hF.data(), dOutF, sizeof(float) * hF.size(), hipMemcpyDeviceToHost, s)) != hipSuccess ||
             (e = hipMemcpyAsync(hT.data(), dTot, sizeof(double) * nu, hipMemcpyDeviceToHost, s)) != hipSuccess ||
             (e = hipMemcpyAsync(hD.data(), dOutD, sizeof(double) * hD.size(), hipMemcpyDeviceToHost, s)) != hipSuccess ||
             (e = hipStreamSynchronize(s)) != hipSuccess) { htkamd_set_error("decoder_run: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP; }
      } else (void)hipStreamSynchronize(s);
      if (rc) return rc;
      { float ms = 0.0f;
        if (hipEventElapsedTime(&ms, d->ev[0], d->ev[1]) == hipSuccess) d->lastScoreMs += ms;
        if (hipEventElapsedTime(&ms, d->ev[2], d->ev[3]) == hipSuccess) d->lastTokenMs += ms; }
      for (int k = 0; k < nu; k++) { d->lastLive[0] += (long long)hLive[2 * k]; d->lastLive[1] += (long long)hLive[2 * k + 1]; }
#ifdef DEC_NEED
      fprintf(stderr, "k_decode: %llu (frame, tied state) scores asked for by live tokens, of %d x %lld in the dense block\n", hLive[2 * (size_t)nu + 6], ns, (long long)(score / (ns ? ns : 1)));
#endif
#ifdef DEC_CLK
      fprintf(stderr, "k_decode phase cycles (sum over %d utterances): prune %llu | models %llu | beam tops %llu | fused words %llu | levels %llu | entries %llu\n", nu,
              hLive[2 * (size_t)nu], hLive[2 * (size_t)nu + 1], hLive[2 * (size_t)nu + 2], hLive[2 * (size_t)nu + 3], hLive[2 * (size_t)nu + 4], hLive[2 * (size_t)nu + 5]);
#endif
      for (int k = 0; k < nu; k++) {
         nWords[u0 + k] = hI[k]; total[u0 + k] = hT[k];
         if (finalLm) finalLm[u0 + k] = hF[(size_t)nu * maxWords * 3 + k];
         const size_t o = (size_t)(u0 + k) * maxWords, si = (size_t)k * maxWords;
         for (int w = 0; w < maxWords; w++) {
            wordPron[o + w] = hI[nu + si + w]; wordStart[o + w] = hI[nu + (size_t)nu * maxWords + si + w];
            wordEnd[o + w] = hI[nu + (size_t)nu * maxWords * 2 + si + w]; wordScore[o + w] = hF[si + w];
            if (wordLm) wordLm[o + w] = hF[(size_t)nu * maxWords + si + w];
            if (wordAc) wordAc[o + w] = hF[(size_t)nu * maxWords * 2 + si + w];
            if (wordLike) wordLike[o + w] = hD[si + w];
         }
      }
      u0 = u1;
   }
   return HTKAMD_OK;
}
