// decode_n.hip -- K7n: N-best token passing and lattice generation over a recognition network (HVite -n N [-z ext]).
//
// Reference semantics (HRec.c with nToks > 1; oracle/orc_decode_n.c is the restatement this kernel is tested against):
//   TokenSet :100 = the state's best token + up to N-1 RelTokens (float likelihood relative to it, lm, path), sorted, distinct in the
//   word-end node their path ends in; TokSetMerge :279 wherever the 1-best code takes a maximum (StepHMM1 :642, StepHMM2 :790,
//   SetEntryState :1303) with the cut nThresh = genMax - nBeam (:2002); StepWord2 :1046 opens a Path record for the best token and a
//   NxtPath per alternative; CompleteRecognition :2054 -> CreateLattice :1679: MarkPaths numbers the reachable Path records depth
//   first, LatFromPaths :1512 makes every Path / NxtPath an arc (aclike, lmlike, prlike).
//
// Mapping: decode.hip's (one 1024-thread workgroup per utterance, every node PULLS from its predecessors in the order the reference's
// senders are stepped, emitting models in registers, zero-time nodes level by level), with token SETS where it has tokens:
//   * a set is a 128-byte record in global memory (L2-resident per utterance); a thread merges into a register-resident copy with the
//     reference's algorithm operand for operand -- the relative likelihoods are floats re-based at every merge, so the order of the
//     merges is the reference's order and the arcs come out with the reference's values;
//   * the state sets of the emitting models are double-buffered per frame (state j of the new column reads several states of the old);
//   * nodes with a large fan-in (loop / back-off null nodes): every thread merges a CONTIGUOUS run of predecessors in order into a
//     partial set in LDS, thread 0 merges the non-empty partial sets in run order -- the sets are the reference's, the re-basing of
//     the relative likelihoods is associated differently (last-bit differences of alternatives' scores are possible there);
//   * Path records: the dense [frame][word node] table of decode.hip plus N-1 alternative slots per entry;
//   * the lattice is built on the device by one thread (depth-first over the table with an explicit stack) and only its nodes and
//     arcs go back to the host.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "decode.h"
#include "decode_ord.h"

#define NT 8                        /* capacity of a token set (HVite -n up to 8) */

struct __attribute__((aligned(16))) TSet {
   double like; float lm; int path;    // the best token
   int n, align;                       // align: Token.align (an index into the utterance's Align records, -1 = NULL; k_decode_ord_n with -m / -f)
   float rl[NT], rlm[NT]; int rp[NT];  // relative tokens; [0] mirrors the best (like 0)
   int ra[NT];                         // RelToken.align (-DPHNALG)
};
struct __attribute__((aligned(8))) AlignRec { int node, state, frame, prev; double like; };      // Align (HRec.c:150-164)

struct NArgs {
   DecNet net;
   const DecUtt *utt; int nUtt;
   const float *score; int ns;         // frame-major: score[score0 + (t-1)*ns + slot]
   TSet *tokA, *tokB;                  // [sum nTok] state sets, double-buffered
   TSet *ex; double *imax;             // [sum nNodes]
   int *pathPrev; double *pathLike; float *pathLm;             // [sum (T+1)*nWordNodes]
   int *altN, *altPrev; double *altLike; float *altLm;         // alternatives: altN[path], others [path*(NT-1) + k]
   int *mark;                                                  // [paths] MarkPaths' usage numbers
   int *stack;                                                 // [nUtt * 2*maxLatNodes] depth-first stack
   int *nodePath;                                              // [nUtt * maxLatNodes] the Path record of a lattice node
   // k_decode_ord_n: the instance list (decode_ord.h) and Path records allocated one by one
   int *seq; int seqCap; int *pos; unsigned char *ooo; int *pathNode, *pathFrame; int pathExtra;
   // ... and alignment records (HVite -n with -m: alignMode & 1 = pri->models, -f: & 2 = pri->states)
   int alignMode; AlignRec *al; int alCap; int *alCount;          // [nUtt * alCap] records, [nUtt] records made
   int *pathAlign, *altAlign;                                     // Path.align [paths], NxtPath.align [paths * (NT-1)]
   int maxAlign; int *arcAlignOff, *alState, *alNode, *alDur; float *alLike;      // lAlign of the arcs: [nUtt * (maxLatArcs + 1)], [nUtt * maxAlign]
   float genBeam, wordBeam, nBeam, lmScale, wordPen, prScale;
   int nToks, maxActive;
   int maxLatNodes, maxLatArcs;
   size_t *pathBase;                   // unused
   int *latN;                          // [nUtt*2] nodes, arcs (or -1 / -3)
   int *nodeFrame, *nodeNet;           // [nUtt*maxLatNodes]
   double *nodeLike;
   int *arcStart, *arcEnd; float *arcAc, *arcLm, *arcPr; double *arcScore;     // [nUtt*maxLatArcs]
   double *total;
};

__device__ __forceinline__ void ts_null(TSet &s) { s.like = LZERO; s.lm = 0.0f; s.path = -1; s.n = 1; s.align = -1; s.rl[0] = 0.0f; s.rlm[0] = 0.0f; s.rp[0] = -1; s.ra[0] = -1; }
// the word-end node a path ends in (TokSetMerge compares path->node): the dense table's column, or -- Path records allocated one by one
// (k_decode_ord_n) -- the record's node
struct KeyOf {
   const int *pathNode; int nW;
   __device__ __forceinline__ int operator()(int path) const { return path < 0 ? -1 : (pathNode ? pathNode[path] : path % nW); }
};

// TokSetMerge (HRec.c:279): token (cLike, cLm, cPath) with the relative tokens of `src` merged into `res`
__device__ void ts_merge(TSet &res, double cLike, float cLm, int cPath, const TSet &src, float nThresh, int nToks, const KeyOf key_of, const int cAlign = -1)
{
   float tl[NT], tlm[NT]; int tp[NT], ta[NT]; int tn; double tLike;
   if (cLike >= res.like) {
      if (!(cLike > nThresh)) return;
      if (res.like > nThresh) {                            // exchange
         tLike = res.like; tn = res.n;
         for (int k = 0; k < res.n; k++) { tl[k] = res.rl[k]; tlm[k] = res.rlm[k]; tp[k] = res.rp[k]; ta[k] = res.ra[k]; }
         res.like = cLike; res.lm = cLm; res.path = cPath; res.align = cAlign; res.n = src.n;
         for (int k = 0; k < src.n; k++) { res.rl[k] = src.rl[k]; res.rlm[k] = src.rlm[k]; res.rp[k] = src.rp[k]; res.ra[k] = src.ra[k]; }
      } else {
         res.like = cLike; res.lm = cLm; res.path = cPath; res.align = cAlign; res.n = src.n;
         for (int k = 0; k < src.n; k++) { res.rl[k] = src.rl[k]; res.rlm[k] = src.rlm[k]; res.rp[k] = src.rp[k]; res.ra[k] = src.ra[k]; }
         return;
      }
   } else {
      if (!(cLike > nThresh)) return;
      tLike = cLike; tn = src.n;
      for (int k = 0; k < src.n; k++) { tl[k] = src.rl[k]; tlm[k] = src.rlm[k]; tp[k] = src.rp[k]; ta[k] = src.ra[k]; }
   }
   const float diff = (float)(res.like - tLike);
   const float limit = (float)((double)nThresh - tLike);
   for (int i = 0; i < tn; i++) {
      if (tl[i] < limit) break;
      const int key = key_of(tp[i]);
      const float like = tl[i] - diff;
      int mch = -1;
      for (int k = 0; k < res.n; k++) if (key_of(res.rp[k]) == key) { mch = k; break; }
      if (mch < 0) {
         if (res.n < nToks) { mch = res.n++; res.rl[mch] = (float)LZERO; res.rlm[mch] = 0.0f; res.rp[mch] = -1; res.ra[mch] = -1; }
         else mch = res.n - 1;
      }
      if (like > res.rl[mch]) {
         for (mch--; mch >= 0 && like > res.rl[mch]; mch--) { res.rl[mch + 1] = res.rl[mch]; res.rlm[mch + 1] = res.rlm[mch]; res.rp[mch + 1] = res.rp[mch]; res.ra[mch + 1] = res.ra[mch]; }
         mch++;
         res.rp[mch] = tp[i]; res.rlm[mch] = tlm[i]; res.ra[mch] = ta[i]; res.rl[mch] = like;
      }
   }
}

// predecessors k0..k1 (step 1) of a node merged in order into `res` (SetEntryState over StepInst2's sends)
__device__ void pull_sets(const NArgs &a, const TSet *ex, int k0, int k1, float gT, float wT, float nT, TSet &res)
{
   for (int k = k0; k < k1; k++) {
      const int ps = a.net.predSrc[k];
      const float lm = a.net.predLike[k];
      const TSet &e = ex[ps & 0x7fffffff];
      if (!(e.like > gT)) continue;
      if (ps < 0 && e.like < wT) continue;
      const double c = e.like + lm * a.lmScale;
      if (!(c > gT)) continue;
      TSet x = e;
      for (int q = 0; q < x.n; q++) x.rlm[q] = e.rlm[q] + lm;
      ts_merge(res, c, e.lm + lm, e.path, x, nT, a.nToks, KeyOf{nullptr, a.net.nWordNodes});
   }
}

__device__ __forceinline__ double block_max_n(double v, double *red)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = (w > v) ? w : v; }
   const int wv = threadIdx.x >> 6;
   __syncthreads();
   if ((threadIdx.x & 63) == 0) red[wv] = v;
   __syncthreads();
   double r = red[0];
   for (int i = 1; i < DEC_THREADS / 64; i++) r = (red[i] > r) ? red[i] : r;
   return r;
}

// StepWord2 for the token set `st` that entered word node n at frame t: the Path record and the exit set
__device__ void word_exit(const NArgs &a, const DecUtt &ud, int n, int t, const TSet &st, TSet &e)
{
   const DecNet &N = a.net;
   e = st;
   e.like += a.wordPen;
   e.like += N.pronProb[n] * a.prScale;
   const size_t pid = (size_t)t * N.nWordNodes + N.wordIdx[n];
   a.pathPrev[ud.path0 + pid] = st.path; a.pathLike[ud.path0 + pid] = e.like; a.pathLm[ud.path0 + pid] = e.lm;
   a.altN[ud.path0 + pid] = st.n - 1;
   for (int k = 1; k < st.n; k++) {
      const size_t z = (ud.path0 + pid) * (NT - 1) + (k - 1);
      a.altLike[z] = e.like + st.rl[k]; a.altLm[z] = st.rlm[k]; a.altPrev[z] = st.rp[k];
   }
   e.path = (int)pid; e.lm = 0.0f;
   e.n = 1; e.rl[0] = 0.0f; e.rlm[0] = 0.0f; e.rp[0] = e.path;
}

// lAlign of one arc (LatFromPaths HRec.c:1582-1656 with -DPHNALG): the arc's chain of records, latest first; a state record's likelihood
// is the running difference, a model record closes the model met before it.  Returns the number of records, -1 if they do not fit.
__device__ int emit_lalign(const NArgs &a, int sel, size_t outBase, int at, int a0, int pathFrame, double plike, float plm, double wp, int prevFrame, double prevLike, bool hasPrev)
{
   const AlignRec *AL = a.al + (size_t)sel * a.alCap;
   int cnt = 0;
   for (int al = a0; al >= 0; al = AL[al].prev) cnt++;
   if (cnt == 0) return 0;
   if (at + cnt > a.maxAlign) return -1;
   int i = cnt, frame = pathFrame, prr = -1, labpr = -1;
   double like = plike - plm * a.lmScale - wp;
   for (int al = a0; al >= 0; al = AL[al].prev) {
      const AlignRec A = AL[al];
      int durF, labNode;
      if (A.state < 0) {
         if (prr < 0) { prr = al; labpr = A.node; continue; }
         durF = AL[prr].frame - A.frame;
         like = AL[prr].like - A.like;
         prr = al;
         labNode = labpr; labpr = A.node;
      } else {
         labNode = A.node;
         durF = frame - A.frame;
         like = like - A.like;
         frame = A.frame;
      }
      i--;
      a.alState[outBase + at + i] = A.state; a.alNode[outBase + at + i] = labNode; a.alDur[outBase + at + i] = durF; a.alLike[outBase + at + i] = (float)like;
      like = A.like;
   }
   if (prr >= 0) {
      int durF;
      if (hasPrev) { durF = AL[prr].frame - prevFrame; like = AL[prr].like - prevLike; }
      else { durF = AL[prr].frame; like = AL[prr].like; }
      i--;
      a.alState[outBase + at + i] = -1; a.alNode[outBase + at + i] = labpr; a.alDur[outBase + at + i] = durF; a.alLike[outBase + at + i] = (float)like;
   }
   return cnt;
}

// CompleteRecognition (HRec.c:2054) -> CreateLattice (:1679): MarkPaths (:1664) numbers the Path records reachable from the final token
// set depth first, LatFromPaths (:1512) makes every Path / NxtPath an arc.  By ONE thread.  Where a Path record lies: the dense
// [frame][word node] table of k_decode_n, or records allocated one by one with their frame and node beside them (k_decode_ord_n).
struct PathView {
   const int *pathFrame, *pathNode; int nW; const int *wordNode;
   __device__ __forceinline__ int frame(int p) const { return pathFrame ? pathFrame[p] : p / nW; }
   __device__ __forceinline__ int node(int p) const { return pathNode ? pathNode[p] : wordNode[p % nW]; }
};

__device__ void build_lattice(const NArgs &a, const DecUtt &ud, int u, const TSet &fin, const PathView pv, const int sel = 0)
{
   const DecNet &N = a.net;
   const int T = ud.T;
   a.total[u] = LZERO;
   int *latN = a.latN + 2 * u;
   latN[0] = -1; latN[1] = 0;
   if (fin.path < 0) return;
   a.total[u] = fin.like;
   const size_t nb = (size_t)u * a.maxLatNodes, ab = (size_t)u * a.maxLatArcs;
   int *stk = a.stack + (size_t)u * 2 * a.maxLatNodes;
   int *nodePath = a.nodePath + nb;
   int nn = 1, nl = 0, sp = 0;
   bool overflow = false;
   // the root (a Path that is not in the table): node 1; children = fin.path, then fin.rp[1..]
   nn = 2; nl = 1;
   a.nodeFrame[nb + 0] = 0; a.nodeNet[nb + 0] = -1; a.nodeLike[nb + 0] = 0.0;
   a.nodeFrame[nb + 1] = T; a.nodeNet[nb + 1] = -2; a.nodeLike[nb + 1] = fin.like;
   // visit(p): number it, push it; children are looked at in order prev, alt 0, alt 1, ...
#define VISIT(p_) do { const int pp_ = (p_); if (pp_ >= 0 && a.mark[ud.path0 + pp_] == 0) { \
      if (nn >= a.maxLatNodes) overflow = true; else { a.mark[ud.path0 + pp_] = nn; nodePath[nn] = pp_; \
         a.nodeFrame[nb + nn] = pv.frame(pp_); a.nodeNet[nb + nn] = pv.node(pp_); a.nodeLike[nb + nn] = a.pathLike[ud.path0 + pp_]; nn++; nl++; \
         stk[2 * sp] = pp_; stk[2 * sp + 1] = 0; sp++; } } } while (0)
   for (int c = 0; c < fin.n && !overflow; c++) {
      if (c > 0) nl++;
      VISIT(c == 0 ? fin.path : fin.rp[c]);
      while (sp > 0 && !overflow) {
         const int p = stk[2 * (sp - 1)];
         const int ch = stk[2 * (sp - 1) + 1]++;
         const int nAlt = a.altN[ud.path0 + p];
         if (ch == 0) VISIT(a.pathPrev[ud.path0 + p]);
         else if (ch - 1 < nAlt) { nl++; VISIT(a.altPrev[(ud.path0 + p) * (NT - 1) + (ch - 1)]); }
         else sp--;
      }
   }
#undef VISIT
   if (overflow || nl > a.maxLatArcs) { latN[0] = -3; return; }
   int ln = 0, nAl = 0;
   const size_t aob = (size_t)u * (a.maxLatArcs + 1), alb = (size_t)u * a.maxAlign;
   // arcs of the root
   for (int c = 0; c < fin.n; c++) {
      const int prev = (c == 0) ? fin.path : fin.rp[c];
      const double plike = (c == 0) ? fin.like : fin.like + fin.rl[c];
      const float plm = (c == 0) ? fin.lm : fin.rlm[c];
      const double prlk = (prev >= 0) ? a.pathLike[ud.path0 + prev] : 0.0;
      a.arcStart[ab + ln] = (prev >= 0) ? a.mark[ud.path0 + prev] : 0; a.arcEnd[ab + ln] = 1;
      a.arcAc[ab + ln] = (float)(plike - prlk - plm * a.lmScale - 0.0); a.arcLm[ab + ln] = plm; a.arcPr[ab + ln] = 0.0f; a.arcScore[ab + ln] = plike;
      if (a.alignMode) {
         a.arcAlignOff[aob + ln] = nAl;
         const int r = emit_lalign(a, sel, alb, nAl, (c == 0) ? fin.align : fin.ra[c], T, plike, plm, 0.0, (prev >= 0) ? pv.frame(prev) : 0, prlk, prev >= 0);
         if (r < 0) { latN[0] = -3; return; }
         nAl += r;
      }
      ln++;
   }
   for (int i = 2; i < nn; i++) {
      const int p = nodePath[i];
      const int node = a.nodeNet[nb + i];
      const int nAlt = a.altN[ud.path0 + p];
      for (int c = 0; c <= nAlt; c++) {
         const size_t z = (ud.path0 + p) * (NT - 1) + (c - 1);
         const int prev = (c == 0) ? a.pathPrev[ud.path0 + p] : a.altPrev[z];
         const double plike = (c == 0) ? a.pathLike[ud.path0 + p] : a.altLike[z];
         const float plm = (c == 0) ? a.pathLm[ud.path0 + p] : a.altLm[z];
         const double prlk = (prev >= 0) ? a.pathLike[ud.path0 + prev] : 0.0;
         const double wp = a.wordPen;
         float ac = (float)(plike - prlk - plm * a.lmScale - wp);
         const float pr = N.pronProb[node];
         ac -= pr * a.prScale;
         a.arcStart[ab + ln] = (prev >= 0) ? a.mark[ud.path0 + prev] : 0; a.arcEnd[ab + ln] = i;
         a.arcAc[ab + ln] = ac; a.arcLm[ab + ln] = plm; a.arcPr[ab + ln] = pr; a.arcScore[ab + ln] = plike;
         if (a.alignMode) {
            a.arcAlignOff[aob + ln] = nAl;
            const int r = emit_lalign(a, sel, alb, nAl, (c == 0) ? a.pathAlign[ud.path0 + p] : a.altAlign[z], pv.frame(p), plike, plm, wp, (prev >= 0) ? pv.frame(prev) : 0, prlk, prev >= 0);
            if (r < 0) { latN[0] = -3; return; }
            nAl += r;
         }
         ln++;
      }
   }
   if (a.alignMode) a.arcAlignOff[aob + ln] = nAl;
   latN[0] = nn; latN[1] = ln;
}

__global__ __launch_bounds__(DEC_THREADS) void k_decode_n(NArgs a)
{
   __shared__ double red[DEC_THREADS / 64];
   __shared__ double red2[DEC_THREADS / 64];
   __shared__ float thr[3];
   __shared__ unsigned int usel[3], uhist[256];
   extern __shared__ unsigned char dynLds[];           // partial sets of a wide node: DEC_THREADS x TSet
   TSet *part = (TSet *)dynLds;
   const int u = blockIdx.x, tid = threadIdx.x;
   if (u >= a.nUtt) return;
   const DecUtt ud = a.utt[u];
   const DecNet &N = a.net;
   const int T = ud.T, nW = N.nWordNodes;
   TSet *cur = a.tokA + ud.tok0, *nxt = a.tokB + ud.tok0, *ex = a.ex + ud.node0;
   double *imax = a.imax + ud.node0;
   const float *tpBase = N.transP;

   { TSet z; ts_null(z);
     for (int i = tid; i < N.nTok; i += DEC_THREADS) { cur[i] = z; nxt[i] = z; }
     for (int i = tid; i < N.nNodes; i += DEC_THREADS) { ex[i] = z; imax[i] = LZERO; } }
   for (size_t i = tid; i < (size_t)(T + 1) * nW; i += DEC_THREADS) a.mark[ud.path0 + i] = 0;
   if (tid == 0) { thr[0] = (float)LSMALL; thr[1] = (float)LSMALL; thr[2] = (float)LSMALL; }
   __syncthreads();

   for (int t = 0; t <= T; t++) {
      if (t >= 1 && a.maxActive > 0) {
         // ---- maximum-model pruning (ProcessObservation HRec.c:1966-1985), as in decode.hip: when more than maxActive instances are
         // attached, those whose max (a float) lies below the (maxActive+1)-th largest lose every token set before pass 1
         const float gTp = thr[0];
         if (tid == 0) usel[0] = 0;
         __syncthreads();
         int cnt = 0;
         for (int n = tid; n < N.nNodes; n += DEC_THREADS) { const double v = imax[n]; if (v >= gTp && v > LSMALL) cnt++; }
         if (cnt) atomicAdd(&usel[0], (unsigned)cnt);
         __syncthreads();
         const int nact = (int)usel[0];
         if (nact > a.maxActive) {
            if (tid == 0) { usel[1] = 0; usel[2] = (unsigned)a.maxActive; }
            unsigned int mask = 0;
            for (int pass = 0; pass < 4; pass++) {
               const int shift = 24 - 8 * pass;
               for (int i = tid; i < 256; i += DEC_THREADS) uhist[i] = 0;
               __syncthreads();
               const unsigned int prefix = usel[1];
               for (int n = tid; n < N.nNodes; n += DEC_THREADS) {
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
            if (uth > (float)LSMALL)
               for (int n = tid; n < N.nNodes; n += DEC_THREADS) {
                  const double v = imax[n];
                  if (!(v >= gTp && v > LSMALL) || !(v < (double)uth)) continue;
                  TSet z; ts_null(z);
                  imax[n] = LZERO; ex[n] = z;
                  const int4 ni = N.nodeInfo[n];
                  const int nt = ((ni.x & 15) == HTKAMD_NODE_HMM) ? ((ni.x >> 4) & 255) - 1 : 1;
                  for (int i = 0; i < nt; i++) cur[ni.y + i] = z;
               }
            __syncthreads();
         }
      }
      if (t >= 1) {
         const float gT = thr[0], nT = thr[2];             // thresholds of the previous frame
         double myGen = LZERO, myWord = LZERO;
         for (int hk = tid; hk < N.nHmm; hk += DEC_THREADS) {
            const int n = N.hmmNodes[hk];
            const int4 ni = N.nodeInfo[n];
            const int NS = (ni.x >> 4) & 255, t0 = ni.y;
            const float *tp = tpBase + ni.z;
            const bool detached = imax[n] < gT;
            bool live = false;
            for (int i = 1; i < NS; i++) {
               if (i > 1 && detached) { TSet z; ts_null(z); cur[t0 + i - 1] = z; }
               if (cur[t0 + i - 1].like > LSMALL) live = true;
            }
            TSet exS; ts_null(exS);
            double mx = LZERO;
            if (live) {
               for (int j = 2; j < NS; j++) {
                  int lo = 1, hi = NS - 1;
                  while (lo < NS && !(tp[(lo - 1) * NS + (j - 1)] > LSMALL)) lo++;
                  while (hi > 1 && !(tp[(hi - 1) * NS + (j - 1)] > LSMALL)) hi--;
                  if (lo > hi) { lo = 1; hi = NS - 1; }
                  TSet res = cur[t0 + lo - 1];
                  res.like += tp[(lo - 1) * NS + (j - 1)];
                  for (int i = lo + 1; i <= hi; i++) {
                     const TSet &si = cur[t0 + i - 1];
                     ts_merge(res, si.like + tp[(i - 1) * NS + (j - 1)], si.lm, si.path, si, nT, a.nToks, KeyOf{nullptr, nW});
                  }
                  if (res.like > gT) {
                     const int st = N.hmmState[ni.w + (j - 2)];
                     res.like += a.score[ud.score0 + (size_t)(t - 1) * a.ns + N.stateSlot[st]];
                     if (res.like > mx) mx = res.like;
                  } else ts_null(res);
                  nxt[t0 + j - 1] = res;
               }
               { TSet z; ts_null(z); nxt[t0] = z; }          // entry consumed
               {
                  int lo = 2, hi = NS - 1;
                  while (lo < NS && !(tp[(lo - 1) * NS + (NS - 1)] > LSMALL)) lo++;
                  while (hi > 1 && !(tp[(hi - 1) * NS + (NS - 1)] > LSMALL)) hi--;
                  if (lo > hi) { lo = 2; hi = NS - 1; }
                  TSet res = nxt[t0 + lo - 1];
                  res.like += tp[(lo - 1) * NS + (NS - 1)];
                  for (int i = lo + 1; i <= hi; i++) {
                     const TSet &si = nxt[t0 + i - 1];
                     ts_merge(res, si.like + tp[(i - 1) * NS + (NS - 1)], si.lm, si.path, si, nT, a.nToks, KeyOf{nullptr, nW});
                  }
                  if (res.like > LSMALL) {
                     exS = res;
                     const double w = res.like + N.wdlk[n];
                     if (w > myWord) myWord = w;
                  }
               }
               if (mx > myGen) myGen = mx;
            } else {
               TSet z; ts_null(z);
               for (int i = 1; i < NS; i++) nxt[t0 + i - 1] = z;
            }
            ex[n] = exS; imax[n] = (double)(float)mx;
         }
         const double genMax = block_max_n(myGen, red);
         const double wordMax = block_max_n(myWord, red2);
         if (tid == 0) {
            float w = (float)(wordMax - a.wordBeam); if (w < (float)LSMALL) w = (float)LSMALL;
            float g = (float)(genMax - a.genBeam); if (g < (float)LSMALL) g = (float)LSMALL;
            float nn = (float)(genMax - a.nBeam); if (nn < (float)(LSMALL / 2)) nn = (float)(LSMALL / 2);
            thr[0] = g; thr[1] = w; thr[2] = nn;
         }
         __syncthreads();
         { TSet *sw = cur; cur = nxt; nxt = sw; }            // the new column is the current one from here on
      }
      const float gT = thr[0], wT = thr[1], nT = thr[2];
      for (int L = 0; L < N.nLevels; L++) {
         const int l0 = N.levelOffAll[L], lw = N.levelWideAll[L], l1 = N.levelOffAll[L + 1];
         for (int k = l0 + tid; k < lw; k += DEC_THREADS) {
            const int n = N.levelNodesAll[k];
            const int4 ni = N.nodeInfo[n];
            const int kind = ni.x & 15;
            TSet st; ts_null(st);
            pull_sets(a, ex, N.predOff[n], N.predOff[n + 1], gT, wT, nT, st);
            if (t == 0 && n == N.initial) { st.like = 0.0; st.lm = 0.0f; st.path = -1; st.n = 1; st.rl[0] = 0.0f; st.rlm[0] = 0.0f; st.rp[0] = -1; }
            TSet e; ts_null(e);
            if (kind == HTKAMD_NODE_HMM) {                  // tee model: StepHMM2
               const int NS = (ni.x >> 4) & 255;
               cur[ni.y] = st;
               e = ex[n];
               const double m2 = (st.like > imax[n]) ? (double)(float)st.like : imax[n];
               if (t >= 1 && m2 < gT) { ts_null(e); imax[n] = LZERO; }
               else {
                  imax[n] = m2;
                  if (st.like > LSMALL) ts_merge(e, st.like + tpBase[ni.z + (NS - 1)], st.lm, st.path, st, nT, a.nToks, KeyOf{nullptr, nW});
               }
            } else if (!(st.like > LSMALL)) imax[n] = LZERO;
            else {
               imax[n] = (double)(float)st.like;
               if (kind == HTKAMD_NODE_WORD) word_exit(a, ud, n, t, st, e);
               else e = st;
            }
            ex[n] = e;
         }
         for (int k = lw; k < l1; k++) {                   // wide fan-in: contiguous runs per thread, then thread 0 in run order
            const int n = N.levelNodesAll[k];
            const int p0 = N.predOff[n], p1 = N.predOff[n + 1];
            const int per = (p1 - p0 + DEC_THREADS - 1) / DEC_THREADS;
            TSet mine; ts_null(mine);
            const int b0 = p0 + tid * per, b1 = (b0 + per < p1) ? b0 + per : p1;
            if (b0 < p1) pull_sets(a, ex, b0, b1, gT, wT, nT, mine);
            part[tid] = mine;
            __syncthreads();
            if (tid == 0) {
               TSet st; ts_null(st);
               for (int q = 0; q < DEC_THREADS; q++) {
                  const TSet &pq = part[q];
                  if (!(pq.like > LSMALL)) continue;
                  ts_merge(st, pq.like, pq.lm, pq.path, pq, nT, a.nToks, KeyOf{nullptr, nW});
               }
               if (t == 0 && n == N.initial) { st.like = 0.0; st.lm = 0.0f; st.path = -1; st.n = 1; st.rl[0] = 0.0f; st.rlm[0] = 0.0f; st.rp[0] = -1; }
               TSet e; ts_null(e);
               imax[n] = (st.like > LSMALL) ? (double)(float)st.like : LZERO;
               if (st.like > LSMALL) {
                  if (N.kind[n] == HTKAMD_NODE_WORD) word_exit(a, ud, n, t, st, e);
                  else e = st;
               }
               ex[n] = e;
            }
            __syncthreads();
         }
         __syncthreads();
      }
      if (t < T) {
         for (int hk = tid; hk < N.nHmm; hk += DEC_THREADS) {
            const int n = N.hmmNodes[hk];
            const int4 ni = N.nodeInfo[n];
            if ((ni.x >> 12) & 1) continue;
            TSet en; ts_null(en);
            pull_sets(a, ex, N.predOff[n], N.predOff[n + 1], gT, wT, nT, en);
            cur[ni.y] = en;
            if (en.like > imax[n]) imax[n] = (double)(float)en.like;
         }
         __syncthreads();
      }
   }

   // ---- CompleteRecognition -> CreateLattice
   if (tid == 0) build_lattice(a, ud, u, ex[N.final], PathView{nullptr, nullptr, nW, N.wordNode});
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_decode_ord_n -- N-best token passing in the order of HRec's instance list (decode_ord.hip has the design; this is its token-set
// form).  In N-best mode the list's order decides more than exact ties: every TokSetMerge re-bases its relative tokens as floats
// (HRec.c:361-364), so the order in which a node's senders are stepped is in the last bit of every alternative's likelihood -- and in
// which of two alternatives of nearly equal likelihood survives.  The run therefore always walks the list.  Pass 1 = k_decode_n's
// StepHMM1 over the list's instances; pass 2 = the walk of k_decode_ord with token sets pushed along the links in link order, each
// merge by one lane on the set in global memory; StepWord2 allocates its Path record (+ NxtPaths) per call.
// NewNRefAlign (HRec.c:598-622): records are allocated one by one per utterance (any thread; their numbers are not part of any result)
__device__ __forceinline__ int new_align(const NArgs &a, int sel, int node, int state, double like, int frame, int prev)
{
   const int i = atomicAdd(&a.alCount[sel], 1);
   if (i >= a.alCap) return -1;                          // (counted: the utterance ends as "did not fit")
   AlignRec r; r.node = node; r.state = state; r.frame = frame; r.prev = prev; r.like = like;
   a.al[(size_t)sel * a.alCap + i] = r;
   return i;
}

__device__ void word_exit_ord(const NArgs &a, const DecUtt &ud, int n, int t, const TSet &st, TSet &e, int pid, int *pathNode, int *pathFrame)
{
   const DecNet &N = a.net;
   e = st;
   e.like += a.wordPen;
   e.like += N.pronProb[n] * a.prScale;
   a.pathPrev[ud.path0 + pid] = st.path; a.pathLike[ud.path0 + pid] = e.like; a.pathLm[ud.path0 + pid] = e.lm;
   pathNode[pid] = n; pathFrame[pid] = t;
   a.altN[ud.path0 + pid] = st.n - 1;
   if (a.alignMode) a.pathAlign[ud.path0 + pid] = st.align;
   for (int k = 1; k < st.n; k++) {
      const size_t z = (ud.path0 + pid) * (NT - 1) + (k - 1);
      a.altLike[z] = e.like + st.rl[k]; a.altLm[z] = st.rlm[k]; a.altPrev[z] = st.rp[k];
      if (a.alignMode) a.altAlign[z] = st.ra[k];
   }
   e.path = pid; e.lm = 0.0f; e.align = -1;
   e.n = 1; e.rp[0] = e.path;                            // rl[0] / rlm[0] / ra[0] stay what the exit set held (AttachInst's rmax: 0, 0, NULL)
}

__global__ __launch_bounds__(ORD_THREADS) void k_decode_ord_n(NArgs a)
{
   __shared__ double red[ORD_THREADS / 64];
   __shared__ double red2[ORD_THREADS / 64];
   __shared__ float thr[3];
   __shared__ unsigned int usel[3], uhist[256];
   __shared__ int scan[ORD_THREADS / 64 + 1];
   __shared__ OrdShared sh;
   const int sel = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   if (sel >= a.nUtt) return;
   const DecUtt ud = a.utt[sel];
   const DecNet &N = a.net;
   const int T = ud.T, u = ud.idx;
   TSet *cur = a.tokA + ud.tok0, *nxt = a.tokB + ud.tok0, *ex = a.ex + ud.node0;
   volatile double *imax = a.imax + ud.node0;
   volatile int *pos = a.pos + ud.node0;
   volatile unsigned char *ooo = a.ooo + ud.node0;
   int *seqA = a.seq + (size_t)sel * 2 * a.seqCap, *seqB = seqA + a.seqCap;
   const size_t pathCap = 3 * ((size_t)(T + 1) * N.nWordNodes) + (size_t)a.pathExtra;
   int *pathNode = a.pathNode + ud.path0, *pathFrame = a.pathFrame + ud.path0;
   const float *tpBase = N.transP;
   const KeyOf ko{pathNode, 0};

   { TSet z; ts_null(z);
     for (int i = tid; i < N.nTok; i += ORD_THREADS) { cur[i] = z; nxt[i] = z; }
     for (int i = tid; i < N.nNodes; i += ORD_THREADS) { ex[i] = z; imax[i] = LZERO; pos[i] = -1; ooo[i] = 0; } }
   for (size_t i = tid; i < pathCap; i += ORD_THREADS) a.mark[ud.path0 + i] = 0;
   if (tid == 0) { thr[0] = (float)LSMALL; thr[1] = (float)LSMALL; thr[2] = (float)LSMALL; sh.tail = 0; sh.nPath = 0; sh.status = 0; sh.base = 0; sh.cn = 0; if (a.alignMode) a.alCount[sel] = 0; }
   __syncthreads();
   OrdCtx c;
   c.N = &N; c.seq = seqA; c.pos = pos; c.ooo = ooo; c.imax = imax; c.seqCap = a.seqCap; c.sh = &sh;
   if (tid == 0) {                                          // StartRecognition (HRec.c:1884)
      o_attach(c, N.initial);
      TSet &s0 = cur[N.nodeInfo[N.initial].y];
      s0.like = 0.0; s0.lm = 0.0f; s0.path = -1; s0.n = 1;
      imax[N.initial] = 0.0;
   }
   __threadfence_block();
   __syncthreads();

   for (int t = 0; t <= T; t++) {
      if (t >= 1) {
         if (a.maxActive > 0 && sh.tail > a.maxActive) {    // maximum-model pruning (HRec.c:1966-1985), as in k_decode_ord
            int cnt = 0;
            if (tid == 0) usel[0] = 0;
            __syncthreads();
            for (int i = tid; i < sh.tail; i += ORD_THREADS) if (c.seq[i] >= 0) cnt++;
            if (cnt) atomicAdd(&usel[0], (unsigned)cnt);
            __syncthreads();
            if ((int)usel[0] > a.maxActive) {
               if (tid == 0) { usel[1] = 0; usel[2] = (unsigned)a.maxActive; }
               unsigned int mask = 0;
               for (int pass = 0; pass < 4; pass++) {
                  const int shift = 24 - 8 * pass;
                  for (int i = tid; i < 256; i += ORD_THREADS) uhist[i] = 0;
                  __syncthreads();
                  const unsigned int prefix = usel[1];
                  for (int i = tid; i < sh.tail; i += ORD_THREADS) {
                     const int n = c.seq[i];
                     if (n < 0) continue;
                     unsigned int k = __float_as_uint((float)imax[n]);
                     k ^= (k >> 31) ? 0xFFFFFFFFu : 0x80000000u;
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
               if (uth > (float)LSMALL)
                  for (int i = tid; i < sh.tail; i += ORD_THREADS) {
                     const int n = c.seq[i];
                     if (n < 0 || !((float)imax[n] < uth)) continue;
                     TSet z; ts_null(z);
                     c.seq[i] = -1; pos[n] = -1; imax[n] = LZERO; ex[n] = z;
                     const int4 ni = N.nodeInfo[n];
                     const int nt = ((ni.x & 15) == HTKAMD_NODE_HMM) ? ((ni.x >> 4) & 255) - 1 : 1;
                     for (int q = 0; q < nt; q++) { cur[ni.y + q] = z; nxt[ni.y + q] = z; }
                  }
            }
            __syncthreads();
         }
         {  // the blanks out of the list
            int *src = (int *)c.seq, *dst = (src == seqA) ? seqB : seqA;
            const int tl = sh.tail;
            int outBase = 0;
            for (int b0 = 0; b0 < tl; b0 += ORD_THREADS) {
               const int i = b0 + tid;
               const int n = (i < tl) ? src[i] : -1;
               const unsigned long long m = __ballot(n >= 0);
               if (lane == 0) scan[wv] = __popcll(m);
               __syncthreads();
               int off = outBase;
               for (int w = 0; w < wv; w++) off += scan[w];
               int tot = 0;
               for (int w = 0; w < ORD_THREADS / 64; w++) tot += scan[w];
               if (n >= 0) { const int o = off + __popcll(m & ((1ull << lane) - 1ull)); dst[o] = n; pos[n] = o; }
               outBase += tot;
               __syncthreads();
            }
            c.seq = dst;
            if (tid == 0) sh.tail = outBase;
            __syncthreads();
         }
         // ---- pass 1 over the list's instances (k_decode_n's StepHMM1; StepWord1 for the rest)
         const float gT = thr[0], nT = thr[2];
         double myGen = LZERO, myWord = LZERO;
         const int nLive = sh.tail;
         for (int li = tid; li < nLive; li += ORD_THREADS) {
            const int n = c.seq[li];
            const int4 ni = N.nodeInfo[n];
            const int NS = (ni.x >> 4) & 255, t0 = ni.y;
            if ((ni.x & 15) != HTKAMD_NODE_HMM) {             // StepWord1 (HRec.c:1038): the sets' relative tokens stay as they are
               TSet z = cur[t0]; z.like = LZERO; z.lm = 0.0f; z.path = -1; z.align = -1; z.n = 1; nxt[t0] = z;
               TSet ze = ex[n]; ze.like = LZERO; ze.lm = 0.0f; ze.path = -1; ze.align = -1; ze.n = 1; ex[n] = ze;
               imax[n] = LZERO;
               continue;
            }
            const float *tp = tpBase + ni.z;
            TSet exS; ts_null(exS);
            double mx = LZERO;
            for (int j = 2; j < NS; j++) {
               int lo = 1, hi = NS - 1;
               while (lo < NS && !(tp[(lo - 1) * NS + (j - 1)] > LSMALL)) lo++;
               while (hi > 1 && !(tp[(hi - 1) * NS + (j - 1)] > LSMALL)) hi--;
               if (lo > hi) { lo = 1; hi = NS - 1; }
               TSet res = cur[t0 + lo - 1];
               res.like += tp[(lo - 1) * NS + (j - 1)];
               for (int i = lo + 1; i <= hi; i++) {
                  const TSet &si = cur[t0 + i - 1];
                  ts_merge(res, si.like + tp[(i - 1) * NS + (j - 1)], si.lm, si.path, si, nT, a.nToks, ko, si.align);
               }
               if (res.like > gT) {
                  const int st = N.hmmState[ni.w + (j - 2)];
                  const float outp = a.score[ud.score0 + (size_t)(t - 1) * a.ns + N.stateSlot[st]];
                  res.like += outp;
                  if (res.like > mx) mx = res.like;
                  if (a.alignMode & 2) {                       // pri->states (HRec.c:680-704, -DPHNALG): a record where a token enters state j
                     const double alk = res.like - outp - res.lm * a.lmScale;
                     const AlignRec *AL = a.al + (size_t)sel * a.alCap;
                     if (res.align < 0 || AL[res.align].state != j || AL[res.align].node != n) { res.align = new_align(a, sel, n, j, alk, t - 1, res.align); res.ra[0] = res.align; }
                     for (int q = 1; q < res.n; q++)
                        if (res.ra[q] < 0 || AL[res.ra[q]].state != j || AL[res.ra[q]].node != n) res.ra[q] = new_align(a, sel, n, j, alk, t - 1, res.ra[q]);
                  }
               } else { res.like = LZERO; res.lm = 0.0f; res.path = -1; res.align = -1; res.n = 1; }
               nxt[t0 + j - 1] = res;
            }
            { TSet z = cur[t0]; z.like = LZERO; z.lm = 0.0f; z.path = -1; z.align = -1; z.n = 1; nxt[t0] = z; }          // entry consumed
            {
               int lo = 2, hi = NS - 1;
               while (lo < NS && !(tp[(lo - 1) * NS + (NS - 1)] > LSMALL)) lo++;
               while (hi > 1 && !(tp[(hi - 1) * NS + (NS - 1)] > LSMALL)) hi--;
               if (lo > hi) { lo = 2; hi = NS - 1; }
               TSet res = nxt[t0 + lo - 1];
               res.like += tp[(lo - 1) * NS + (NS - 1)];
               for (int i = lo + 1; i <= hi; i++) {
                  const TSet &si = nxt[t0 + i - 1];
                  ts_merge(res, si.like + tp[(i - 1) * NS + (NS - 1)], si.lm, si.path, si, nT, a.nToks, ko, si.align);
               }
               if (res.like > LSMALL) {
                  const double w = res.like + N.wdlk[n];
                  if (w > myWord) myWord = w;
                  if ((a.alignMode & 1) && !((ni.x >> 12) & 1)) {      // pri->models, not a tee model (HRec.c:762-776): the model's exit record per token
                     const double alk = res.like - res.lm * a.lmScale;
                     res.align = new_align(a, sel, n, -1, alk, t, res.align); res.ra[0] = res.align;
                     for (int q = 1; q < res.n; q++) res.ra[q] = new_align(a, sel, n, -1, alk, t, res.ra[q]);
                  }
                  exS = res;
               } else { exS = res; exS.like = LZERO; exS.lm = 0.0f; exS.path = -1; exS.align = -1; exS.n = 1; }
            }
            if (mx > myGen) myGen = mx;
            ex[n] = exS; imax[n] = (double)(float)mx;
         }
         const double genMax = o_block_max(myGen, red);
         const double wordMax = o_block_max(myWord, red2);
         if (tid == 0) {
            float w = (float)(wordMax - a.wordBeam); if (w < (float)LSMALL) w = (float)LSMALL;
            float g = (float)(genMax - a.genBeam); if (g < (float)LSMALL) g = (float)LSMALL;
            float nn = (float)(genMax - a.nBeam); if (nn < (float)(LSMALL / 2)) nn = (float)(LSMALL / 2);
            thr[0] = g; thr[1] = w; thr[2] = nn;
         }
         __threadfence_block();
         __syncthreads();
         { TSet *sw = cur; cur = nxt; nxt = sw; }            // the new column is the current one from here on
      }
      // ---- pass 2: wavefront 0 walks the list while it changes
      if (wv == 0) {
         const float gT = thr[0], wT = thr[1], nT = thr[2];
         volatile OrdShared *vs = &sh;
         int idx = 0;
         while (idx < vs->tail && vs->status == 0) {
            const int tl = vs->tail;
            const int cn = (tl - idx < 64) ? tl - idx : 64;
            {
               const int n = (lane < cn) ? c.seq[idx + lane] : -1;
               vs->chunkNode[lane] = n;
               vs->chunkMax[lane] = (n >= 0) ? (float)imax[n] : 0.0f;
               if (lane == 0) { vs->base = idx; vs->cn = cn; }
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = 0; i < cn && vs->status == 0; i++) {
               const int n = vs->chunkNode[i];
               if (n < 0) continue;
               const float nmax = vs->chunkMax[i];
               const int4 ni = N.nodeInfo[n];
               const int kind = ni.x & 15, NS = (ni.x >> 4) & 255, t0 = ni.y;
               if (nmax < gT) {                               // DetachInst
                  if (lane == 0) { TSet z; ts_null(z); o_blank(c, n); pos[n] = -1; imax[n] = LZERO; ex[n] = z; }
                  const int nt = (kind == HTKAMD_NODE_HMM) ? NS - 1 : 1;
                  if (lane < nt) { TSet z; ts_null(z); cur[t0 + lane] = z; nxt[t0 + lane] = z; }
                  __threadfence_block();
                  continue;
               }
               // StepInst2 (HRec.c:1360); lane 0 makes the exit set, then every lane reads it
               if (lane == 0) {
                  if (kind == HTKAMD_NODE_WORD) {
                     const int pid = vs->nPath;
                     if ((size_t)pid >= pathCap) vs->status = -4;
                     else {
                        TSet e = ex[n];
                        const TSet st = cur[t0];
                        const float r0 = e.rl[0], m0 = e.rlm[0]; const int a0 = e.ra[0];
                        word_exit_ord(a, ud, n, t, st, e, pid, pathNode, pathFrame);
                        e.rl[0] = r0; e.rlm[0] = m0; e.ra[0] = a0;
                        vs->nPath = pid + 1;
                        ex[n] = e;
                     }
                  } else if (kind == HTKAMD_NODE_NULL) ex[n] = cur[t0];
                  else if ((ni.x >> 12) & 1) {              // tee model: StepHMM2 (HRec.c:790)
                     const TSet st = cur[t0];
                     TSet e = ex[n];
                     ts_merge(e, st.like + tpBase[ni.z + (NS - 1)], st.lm, st.path, st, nT, a.nToks, ko, st.align);
                     if (a.alignMode & 1) {                   // HRec.c:817-832
                        const double alk = e.like - e.lm * a.lmScale;
                        e.align = new_align(a, sel, n, -1, alk, t, e.align); e.ra[0] = e.align;
                        for (int q = 1; q < e.n; q++) e.ra[q] = new_align(a, sel, n, -1, alk, t, e.ra[q]);
                     }
                     ex[n] = e;
                  }
               }
               __threadfence_block();
               if (vs->status != 0) break;
               const TSet e = ex[n];
               double tkLike = e.like; float tkLm = e.lm;
               if (kind != HTKAMD_NODE_HMM && tkLike < wT) tkLike = LZERO;
               if (tkLike > gT) {
                  const int k0 = N.linkOff[n], k1 = N.linkOff[n + 1];
                  const bool dup = N.dupDest[n] != 0;
                  for (int kb = k0; kb < k1; kb += 64) {
                     const int k = kb + lane;
                     const bool act = k < k1;
                     const int d = act ? N.linkDest[k] : 0;
                     const float lm = act ? N.linkLike[k] : 0.0f;
                     const double xl = tkLike + lm * a.lmScale;
                     const bool pass = act && xl > gT;
                     unsigned long long need = __ballot(pass && pos[d] < 0);
                     while (need) {
                        const int j = __ffsll((long long)need) - 1;
                        need &= need - 1;
                        const int dj = __shfl(d, j);
                        if (lane == 0 && pos[dj] < 0) o_attach(c, dj);
                        __threadfence_block();
                     }
                     if (vs->status != 0) break;
                     unsigned long long todo = __ballot(pass);
                     if (!dup) todo = pass ? (1ull << lane) : 0ull;
                     while (todo) {
                        const int j = dup ? __ffsll((long long)todo) - 1 : lane;
                        todo = dup ? (todo & (todo - 1)) : 0ull;
                        if (lane == j) {                      // SetEntryState (HRec.c:1303): TokSetMerge into the destination's entry set
                           const int td = N.nodeInfo[d].y;
                           TSet x = e;
                           for (int q = 0; q < x.n; q++) x.rlm[q] = e.rlm[q] + lm;
                           TSet res = cur[td];
                           ts_merge(res, xl, tkLm + lm, e.path, x, nT, a.nToks, ko, e.align);
                           cur[td] = res;
                           const double m0 = imax[d];
                           if (res.like > m0) {
                              const float nm = (float)res.like;
                              imax[d] = (double)nm;
                              const int pd = pos[d];
                              if (pd >= vs->base && pd < vs->base + vs->cn) vs->chunkMax[pd - vs->base] = nm;
                           }
                        }
                        if (dup) __threadfence_block();
                     }
                     __threadfence_block();
                  }
               }
            }
            idx += cn;
         }
      }
      __threadfence_block();
      __syncthreads();
      if (sh.status != 0) break;
   }
   if (tid == 0) {
      if (a.alignMode && a.alCount[sel] > a.alCap) sh.status = -4;                 // more alignment records than there was room for
      if (sh.status != 0) { a.total[u] = LZERO; a.latN[2 * u] = sh.status == -4 ? -3 : sh.status; a.latN[2 * u + 1] = 0; }
      else {
         TSet fin; ts_null(fin);
         if (pos[N.final] >= 0) fin = ex[N.final];
         build_lattice(a, ud, u, fin, PathView{pathFrame, pathNode, N.nWordNodes, N.wordNode}, sel);
      }
   }
}

// ------------------------------------------------------------------------------------ host side
static int run_lattice_impl(htkamd_decoder *d, const htkamd_decode_config *cfg, int nToks, float nBeam, const float *dX, const int *frameOff, int nUtt,
                            int maxLatNodes, int maxLatArcs, const htkamd_lattice_out *out, int alignMode, int maxAlign, const htkamd_lattice_align_out *alOut, void *stream);

extern "C" int htkamd_decoder_run_lattice(htkamd_decoder *d, const htkamd_decode_config *cfg, int nToks, float nBeam, const float *dX, const int *frameOff, int nUtt,
                                          int maxLatNodes, int maxLatArcs, const htkamd_lattice_out *out, void *stream)
{
   return run_lattice_impl(d, cfg, nToks, nBeam, dX, frameOff, nUtt, maxLatNodes, maxLatArcs, out, 0, 0, nullptr, stream);
}

extern "C" int htkamd_decoder_run_lattice_align(htkamd_decoder *d, const htkamd_decode_config *cfg, int nToks, float nBeam, int alignMode, const float *dX, const int *frameOff,
                                                int nUtt, int maxLatNodes, int maxLatArcs, int maxAlign, const htkamd_lattice_out *out, const htkamd_lattice_align_out *alOut,
                                                void *stream)
{
   if (alignMode < 0 || alignMode > 3 || (alignMode && (maxAlign < 1 || !alOut || !alOut->arcAlignOff || !alOut->alState || !alOut->alModel || !alOut->alDur || !alOut->alLike))) {
      htkamd_set_error("decoder_run_lattice_align: bad argument"); return HTKAMD_EINVAL;
   }
   return run_lattice_impl(d, cfg, nToks, nBeam, dX, frameOff, nUtt, maxLatNodes, maxLatArcs, out, alignMode, maxAlign, alOut, stream);
}

static int run_lattice_impl(htkamd_decoder *d, const htkamd_decode_config *cfg, int nToks, float nBeam, const float *dX, const int *frameOff, int nUtt,
                            int maxLatNodes, int maxLatArcs, const htkamd_lattice_out *out, int alignMode, int maxAlign, const htkamd_lattice_align_out *alOut, void *stream)
{
   if (!d || !cfg || !frameOff || nUtt < 0 || !out || !out->nNodes || !out->nArcs || maxLatNodes < 2 || maxLatArcs < 1) { htkamd_set_error("decoder_run_lattice: bad argument"); return HTKAMD_EINVAL; }
   if (nToks < 2 || nToks > NT) { htkamd_set_error("decoder_run_lattice: nToks = %d (2..%d tokens per state)", nToks, NT); return HTKAMD_EINVAL; }
   if (nUtt == 0) return HTKAMD_OK;
   hipStream_t s = (hipStream_t)stream;
   htkamd_model *m = d->m;
   const DecNet &N = d->net;
   const int ns = (int)d->usedStates.size();
   const int FR = SCORE_TILE_FRAMES, SL = SCORE_TASK_SLOTS;
   // the list kernel (k_decode_ord_n) unless the static order was asked for: in N-best mode the order of the merges is in every
   // alternative's likelihood, so "exact" is the default here, not a fallback
   int orderMode = d->orderMode;
   if (const char *ev = getenv("HTKAMD_DECODE_ORDER")) orderMode = !strcmp(ev, "fast") ? HTKAMD_ORDER_FAST : !strcmp(ev, "exact") ? HTKAMD_ORDER_EXACT : HTKAMD_ORDER_AUTO;
   if (alignMode && orderMode == HTKAMD_ORDER_FAST) { htkamd_set_error("decoder_run_lattice_align: alignment records need the list kernel (not HTKAMD_ORDER_FAST)"); return HTKAMD_EINVAL; }
   const bool listOrder = orderMode != HTKAMD_ORDER_FAST;
   const size_t pathMul = listOrder ? 3 : 1, pathExtra = listOrder ? 64 : 0;       // StepWord2 "may be repeated" (HRec.c:1046)
   const int seqCap = 8 * N.nNodes + 1024;
   d->lastTied = 0;
   int u0 = 0;
   while (u0 < nUtt) {
      size_t bytes = 0; int u1 = u0;
      while (u1 < nUtt) {
         const size_t T = (size_t)(frameOff[u1 + 1] - frameOff[u1]);
         const size_t b = (size_t)ns * T * 8 + (size_t)N.nTok * 2 * sizeof(TSet) + (size_t)N.nNodes * (sizeof(TSet) + 8) + (pathMul * (T + 1) * (size_t)N.nWordNodes + pathExtra) * (24 + 16 * (NT - 1) + 8) +
                          (listOrder ? (size_t)seqCap * 8 : 0);
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
         ud.score0 = score; ud.tok0 = tok; ud.node0 = node; ud.path0 = path; ud.out0 = 0;
         for (int ti = 0; ti * FR < ud.T; ti++)
            for (int ch = 0; ch * SL < ns; ch++) {
               ScoreTask tk;
               tk.frame0 = ud.frame0 + ti * FR; tk.nFrames = std::min(FR, ud.T - ti * FR);
               tk.slot0 = ch * SL; tk.nSlots = std::min(SL, ns - ch * SL); tk.outSlot0 = ch * SL; tk.ldo = ud.T;
               tk.outBase = ud.score0 + (size_t)ti * FR;
               tasks.push_back(tk);
            }
         score += (size_t)ns * ud.T; tok += (size_t)N.nTok; node += (size_t)N.nNodes; path += pathMul * ((size_t)(ud.T + 1) * N.nWordNodes) + pathExtra;
      }
      int rc = HTKAMD_OK, wsi = 0;
      auto A = [&](size_t n) -> void * {                 // the decoder's own buffers, kept between calls and grown as batches ask (decode.hip)
         const int i = wsi++;
         if (rc) return nullptr;
         if (n < 1) n = 1;
         if (d->wsNCap[i] < n) {
            if (d->wsN[i]) { (void)hipStreamSynchronize(s); (void)hipFree(d->wsN[i]); d->wsN[i] = nullptr; d->wsNCap[i] = 0; }
            const size_t want = n + n / 8;
            hipError_t e = hipMalloc(&d->wsN[i], want);
            if (e != hipSuccess) { htkamd_set_error("decoder_run_lattice: hipMalloc(%zu): %s", want, hipGetErrorString(e)); rc = HTKAMD_ENOMEM; d->wsN[i] = nullptr; return nullptr; }
            d->wsNCap[i] = want;
         }
         return d->wsN[i];
      };
      NArgs a; memset(&a, 0, sizeof(a));
      void *dScore = A(score * 4), *dScoreT = A(score * 4);
      a.tokA = (TSet *)A(tok * sizeof(TSet)); a.tokB = (TSet *)A(tok * sizeof(TSet)); a.ex = (TSet *)A(node * sizeof(TSet)); a.imax = (double *)A(node * 8);
      a.pathPrev = (int *)A(path * 4); a.pathLike = (double *)A(path * 8); a.pathLm = (float *)A(path * 4);
      a.altN = (int *)A(path * 4); a.altPrev = (int *)A(path * 4 * (NT - 1)); a.altLike = (double *)A(path * 8 * (NT - 1)); a.altLm = (float *)A(path * 4 * (NT - 1));
      a.mark = (int *)A(path * 4); a.stack = (int *)A(sizeof(int) * (size_t)nu * 2 * maxLatNodes); a.nodePath = (int *)A(sizeof(int) * (size_t)nu * maxLatNodes);
      void *dUtt = A(sizeof(DecUtt) * nu), *dTasks = A(sizeof(ScoreTask) * tasks.size() + sizeof(int));
      a.latN = (int *)A(sizeof(int) * 2 * nu);
      a.nodeFrame = (int *)A(sizeof(int) * (size_t)nu * maxLatNodes); a.nodeNet = (int *)A(sizeof(int) * (size_t)nu * maxLatNodes); a.nodeLike = (double *)A(8 * (size_t)nu * maxLatNodes);
      a.arcStart = (int *)A(sizeof(int) * (size_t)nu * maxLatArcs); a.arcEnd = (int *)A(sizeof(int) * (size_t)nu * maxLatArcs);
      a.arcAc = (float *)A(4 * (size_t)nu * maxLatArcs); a.arcLm = (float *)A(4 * (size_t)nu * maxLatArcs); a.arcPr = (float *)A(4 * (size_t)nu * maxLatArcs);
      a.arcScore = (double *)A(8 * (size_t)nu * maxLatArcs); a.total = (double *)A(8 * (size_t)nu);
      if (listOrder) {
         a.seq = (int *)A(sizeof(int) * (size_t)nu * 2 * seqCap); a.seqCap = seqCap; a.pos = (int *)A(sizeof(int) * node); a.ooo = (unsigned char *)A(node);
         a.pathNode = (int *)A(path * 4); a.pathFrame = (int *)A(path * 4); a.pathExtra = (int)pathExtra;
      }
      int maxT = 0;
      for (int k = 0; k < nu; k++) maxT = std::max(maxT, utt[k].T);
      if (alignMode) {
         // Align records: one per token of a set where it enters a state (-f) and where it leaves a model (-m), never freed within an
         // utterance (the reference collects garbage; here the utterance must fit): frames x model nodes x tokens x (states + 1), capped
         size_t cap = (size_t)(maxT + 1) * (size_t)N.nHmm * (size_t)nToks * (size_t)((alignMode & 2 ? 3 : 0) + (alignMode & 1 ? 1 : 0));
         if (cap > ((size_t)1 << 27)) cap = (size_t)1 << 27;
         if (cap * (size_t)nu * sizeof(AlignRec) > ((size_t)16 << 30)) cap = ((size_t)16 << 30) / ((size_t)nu * sizeof(AlignRec));
         a.alignMode = alignMode; a.alCap = (int)cap; a.al = (AlignRec *)A(sizeof(AlignRec) * cap * nu); a.alCount = (int *)A(sizeof(int) * nu);
         a.pathAlign = (int *)A(path * 4); a.altAlign = (int *)A(path * 4 * (NT - 1));
         a.maxAlign = maxAlign; a.arcAlignOff = (int *)A(sizeof(int) * (size_t)nu * (maxLatArcs + 1));
         a.alState = (int *)A(sizeof(int) * (size_t)nu * maxAlign); a.alNode = (int *)A(sizeof(int) * (size_t)nu * maxAlign); a.alDur = (int *)A(sizeof(int) * (size_t)nu * maxAlign);
         a.alLike = (float *)A(sizeof(float) * (size_t)nu * maxAlign);
      }
      if (!rc) {
         hipError_t e;
         if ((e = hipMemcpyAsync(dUtt, utt.data(), sizeof(DecUtt) * nu, hipMemcpyHostToDevice, s)) != hipSuccess ||
             (e = hipMemcpyAsync(dTasks, tasks.data(), sizeof(ScoreTask) * tasks.size(), hipMemcpyHostToDevice, s)) != hipSuccess) {
            htkamd_set_error("decoder_run_lattice: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP;
         }
      }
      if (!rc) {
         ScoreArgs sa;
         sa.tasks = (const ScoreTask *)dTasks; sa.nTasks = (int)tasks.size(); sa.X = dX; sa.slotState = d->d_usedStates; sa.out = (float *)dScore;
         sa.stateCompOff = m->d_stateCompOff; sa.compGauss = m->d_compGauss; sa.compLogWt = m->d_compLogWt;
         sa.gparam = m->d_gparam; sa.PS = m->PS; sa.D = m->D; sa.minLogExp = m->minLogExp;
         sa.laddTab = m->d_laddTab; sa.taskCounter = (int *)((char *)dTasks + sizeof(ScoreTask) * tasks.size());
         sa.mfmaTab = m->d_mfmaTab; sa.stateTileOff = m->d_stateTileOff; sa.bf16Tab = m->d_bf16Tab; sa.var = m->d_var;
         sa.NSt = m->NSt; sa.streamWt = m->d_streamWt;
         if (m->NSt > 1 && cfg->scoreMode != HTKAMD_SCORE_EXACT) { htkamd_set_error("decoder_run_lattice: multi-stream sets are scored in the exact mode only"); rc = HTKAMD_EINVAL; }
         else if (m->tiedMix) rc = htkamd_tm_score_block(m, sa, frameOff[u1], m->tmBeam, s);
         else if (cfg->scoreMode != HTKAMD_SCORE_EXACT && cfg->scoreMode != HTKAMD_SCORE_MFMA && cfg->scoreMode != HTKAMD_SCORE_BF16) { htkamd_set_error("decoder_run_lattice: unknown score mode %d", cfg->scoreMode); rc = HTKAMD_EINVAL; }
         else rc = htkamd_launch_score(cfg->scoreMode, m, sa, s);
      }
      if (!rc) rc = htkamd_launch_score_transpose((const float *)dScore, (float *)dScoreT, (const DecUtt *)dUtt, nu, maxT, ns, s);
      if (!rc) {
         a.net = N; a.utt = (const DecUtt *)dUtt; a.nUtt = nu; a.score = (const float *)dScoreT; a.ns = ns;
         a.genBeam = cfg->genBeam; a.wordBeam = cfg->wordBeam; a.nBeam = nBeam; a.lmScale = cfg->lmScale; a.wordPen = cfg->wordPen; a.prScale = cfg->prScale;
         a.nToks = nToks; a.maxActive = cfg->maxActive > 0 ? cfg->maxActive : 0; a.maxLatNodes = maxLatNodes; a.maxLatArcs = maxLatArcs;
         const size_t lds = sizeof(TSet) * DEC_THREADS;
         hipError_t e = hipSuccess;
         if (listOrder) { hipLaunchKernelGGL(k_decode_ord_n, dim3(nu), dim3(ORD_THREADS), 0, s, a); e = hipGetLastError(); d->lastTied += nu; }
         else {
            e = hipFuncSetAttribute((const void *)k_decode_n, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) { hipLaunchKernelGGL(k_decode_n, dim3(nu), dim3(DEC_THREADS), lds, s, a); e = hipGetLastError(); }
         }
         if (e != hipSuccess) { htkamd_set_error("decoder_run_lattice: launch: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP; }
      }
      std::vector<int> hN(2 * nu), hNF, hNN, hAS, hAE; std::vector<double> hNL, hSc, hT(nu); std::vector<float> hAc, hLm, hPr;
      if (!rc) {
         hNF.resize((size_t)nu * maxLatNodes); hNN.resize(hNF.size()); hNL.resize(hNF.size());
         hAS.resize((size_t)nu * maxLatArcs); hAE.resize(hAS.size()); hAc.resize(hAS.size()); hLm.resize(hAS.size()); hPr.resize(hAS.size()); hSc.resize(hAS.size());
         hipError_t e;
#define D2H(h, dptr) ((e = hipMemcpyAsync((h).data(), (dptr), sizeof((h)[0]) * (h).size(), hipMemcpyDeviceToHost, s)) != hipSuccess)
         if (D2H(hN, a.latN) || D2H(hNF, a.nodeFrame) || D2H(hNN, a.nodeNet) || D2H(hNL, a.nodeLike) || D2H(hAS, a.arcStart) || D2H(hAE, a.arcEnd) ||
             D2H(hAc, a.arcAc) || D2H(hLm, a.arcLm) || D2H(hPr, a.arcPr) || D2H(hSc, a.arcScore) || D2H(hT, a.total) ||
             (e = hipStreamSynchronize(s)) != hipSuccess) { htkamd_set_error("decoder_run_lattice: %s", hipGetErrorString(e)); rc = HTKAMD_EHIP; }
#undef D2H
      } else (void)hipStreamSynchronize(s);
      if (rc) return rc;
      if (alignMode) {
         std::vector<int> hOff((size_t)nu * (maxLatArcs + 1)), hSt((size_t)nu * maxAlign), hNd(hSt.size()), hDu(hSt.size()); std::vector<float> hLk(hSt.size());
         hipError_t e;
         if ((e = hipMemcpy(hOff.data(), a.arcAlignOff, sizeof(int) * hOff.size(), hipMemcpyDeviceToHost)) != hipSuccess ||
             (e = hipMemcpy(hSt.data(), a.alState, sizeof(int) * hSt.size(), hipMemcpyDeviceToHost)) != hipSuccess ||
             (e = hipMemcpy(hNd.data(), a.alNode, sizeof(int) * hNd.size(), hipMemcpyDeviceToHost)) != hipSuccess ||
             (e = hipMemcpy(hDu.data(), a.alDur, sizeof(int) * hDu.size(), hipMemcpyDeviceToHost)) != hipSuccess ||
             (e = hipMemcpy(hLk.data(), a.alLike, sizeof(float) * hLk.size(), hipMemcpyDeviceToHost)) != hipSuccess) { htkamd_set_error("decoder_run_lattice_align: %s", hipGetErrorString(e)); return HTKAMD_EHIP; }
         for (int k = 0; k < nu; k++) {
            const int uu = u0 + k;
            const int na = hN[2 * k] > 0 ? hN[2 * k + 1] : 0;
            for (int i = 0; i <= na; i++) alOut->arcAlignOff[(size_t)uu * (maxLatArcs + 1) + i] = hOff[(size_t)k * (maxLatArcs + 1) + i];
            const int nr = na > 0 ? hOff[(size_t)k * (maxLatArcs + 1) + na] : 0;
            for (int i = 0; i < nr; i++) {
               const size_t o = (size_t)uu * maxAlign + i, si = (size_t)k * maxAlign + i;
               alOut->alState[o] = hSt[si]; alOut->alModel[o] = d->hostModel[hNd[si]]; alOut->alDur[o] = hDu[si]; alOut->alLike[o] = hLk[si];
            }
         }
      }
      for (int k = 0; k < nu; k++) {
         const int uu = u0 + k;
         out->nNodes[uu] = hN[2 * k]; out->nArcs[uu] = hN[2 * k + 1];
         if (out->total) out->total[uu] = hT[k];
         const int nn = hN[2 * k] > 0 ? hN[2 * k] : 0, na = hN[2 * k] > 0 ? hN[2 * k + 1] : 0;
         for (int i = 0; i < nn; i++) {
            const size_t o = (size_t)uu * maxLatNodes + i, si = (size_t)k * maxLatNodes + i;
            if (out->nodeFrame) out->nodeFrame[o] = hNF[si];
            if (out->nodeNet) out->nodeNet[o] = hNN[si];
            if (out->nodePron) out->nodePron[o] = hNN[si] >= 0 ? d->hostModel[hNN[si]] : -1;
            if (out->nodeLike) out->nodeLike[o] = hNL[si];
         }
         for (int i = 0; i < na; i++) {
            const size_t o = (size_t)uu * maxLatArcs + i, si = (size_t)k * maxLatArcs + i;
            if (out->arcStart) out->arcStart[o] = hAS[si];
            if (out->arcEnd) out->arcEnd[o] = hAE[si];
            if (out->arcAc) out->arcAc[o] = hAc[si];
            if (out->arcLm) out->arcLm[o] = hLm[si];
            if (out->arcPr) out->arcPr[o] = hPr[si];
            if (out->arcScore) out->arcScore[o] = hSc[si];
         }
      }
      u0 = u1;
   }
   return HTKAMD_OK;
}
