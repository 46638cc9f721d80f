// decode_ord.hip -- K7o: network decoding in HRec's own instance order (the exact-tie path of HVite -w).
//
// Why it exists.  HRec keeps one NetInst per active network node on a list that CHANGES while pass 2 walks it: AttachInst appends
// (HRec.c:1200-1268), ReOrderList / MoveToRecent (:1123-1170) re-append every live successor over zero-time links behind a new
// instance, DetachInst (:1270-1301) unlinks.  SetEntryState (:1303) keeps the FIRST of two tokens of exactly equal likelihood, so which
// of two equally likely histories survives at a node is a function of that list's order -- of the whole activation history of the
// utterance.  k_decode (decode.hip) pulls in a static order that coincides with the list's in all but manufactured cases; it now FLAGS
// an utterance in which two tokens of equal likelihood and different histories met at a node, and such utterances are decoded again
// here, where the list itself is kept on the device and walked as the reference walks it.  (Without a tie every order gives the same
// tokens: the fast kernel's result stands.)
//
// The list as an array.  A node has at most one instance, so an instance is its node; the list is `seq[]` (node numbers in list order,
// -1 where an instance was taken out) with `pos[node]` = the node's place.  AttachInst = append at `tail`; MoveToRecent = blank the
// old place, append; DetachInst = blank.  Pass 2 is one walk over seq[0 .. tail) while tail grows -- what a step attaches or moves
// lies behind the cursor and is reached in the same walk, exactly as `next = pri->nxtInst->link` does (a node moved while it is being
// stepped leaves a blank at the cursor: the walk goes on with what followed it).  Blanks are squeezed out once per frame.
//
// Mapping.  One workgroup of 256 threads per utterance.  Pass 1 (StepHMM1 / StepWord1, beam tops) has no order: all threads, an
// instance each.  Pass 2 is sequential by nature: wavefront 0 walks, 64 list entries fetched at a time; the links of the node being
// stepped are the lanes (the tokens a node sends go to different nodes; attaching, which appends, is taken link by link).  Tokens,
// exit tokens and instance maxima are the arrays k_decode uses; Path records are allocated one by one (StepWord2 may run twice on a node
// in one frame -- "may be repeated", HRec.c:1046 -- and both records can stay referenced).
// Output probabilities come from the same score block (K1, exact mode).  Tested against HVite on the tie files and sweeps
// (tests/test_gpu_decode.py, tests/fuzz_parity.py) and against oracle/orc_decode.c, which walks the same list (oracle/orc_ilist.h).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "decode.h"

#include "decode_ord.h"

__global__ __launch_bounds__(ORD_THREADS) void k_decode_ord(OrdArgs oa)
{
   const DecArgs &a = oa.d;
   __shared__ double red[ORD_THREADS / 64];
   __shared__ double red2[ORD_THREADS / 64];
   __shared__ float thr[2];
   __shared__ float ltp[2048];
   __shared__ int uhist[256];
   __shared__ unsigned int usel[4];
   __shared__ int scan[ORD_THREADS / 64 + 1];
   __shared__ OrdShared sh;
   const int sel = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   if (sel >= a.nUtt) return;
   const DecUtt ud = a.utt[sel];
   const DecNet &N = a.net;
   const int T = ud.T, u = ud.idx;
   Tok *tok = a.tok + ud.tok0, *ex = a.ex + ud.node0;
   volatile double *imax = a.imax + ud.node0;
   volatile int *pos = oa.pos + ud.node0;
   volatile unsigned char *ooo = oa.ooo + ud.node0;
   int *seqA = oa.seq + (size_t)sel * 2 * oa.seqCap, *seqB = seqA + oa.seqCap;
   const size_t pathCap = 3 * ((size_t)(T + 1) * N.nWordNodes) + (size_t)oa.pathExtra;      // as the host laid the records out (decode.hip)
   int *pathPrev = a.pathPrev + ud.path0, *pathNode = oa.pathNode + ud.path0, *pathFrame = oa.pathFrame + ud.path0;
   double *pathLike = a.pathLike + ud.path0; float *pathLm = a.pathLm + ud.path0;
   const bool tpInLds = N.nTpFloats <= 2048;
   if (tpInLds) for (int i = tid; i < N.nTpFloats; i += ORD_THREADS) ltp[i] = N.transP[i];
   const float *tpBase = tpInLds ? ltp : N.transP;

   for (int i = tid; i < N.nTok; i += ORD_THREADS) tok[i] = o_null();
   for (int i = tid; i < N.nNodes; i += ORD_THREADS) { ex[i] = o_null(); imax[i] = LZERO; pos[i] = -1; ooo[i] = 0; }
   if (tid == 0) { thr[0] = (float)LSMALL; thr[1] = (float)LSMALL; sh.tail = 0; sh.nPath = 0; sh.status = 0; sh.base = 0; sh.cn = 0; }
   __syncthreads();
   OrdCtx c;
   c.N = &N; c.seq = seqA; c.pos = pos; c.ooo = ooo; c.imax = imax; c.seqCap = oa.seqCap; c.sh = &sh;
   if (tid == 0) {                                          // StartRecognition (HRec.c:1884): the initial node's instance, a token of likelihood 0
      o_attach(c, N.initial);
      Tok z; z.like = 0.0; z.lm = 0.0f; z.path = -1;
      tok[N.nodeInfo[N.initial].y] = z; imax[N.initial] = 0.0;
   }
   __syncthreads();

   for (int t = 0; t <= T; t++) {
      if (t >= 1) {
         // ---- maximum-model pruning (ProcessObservation HRec.c:1966-1985): more than maxActive instances on the list -> those whose max
         // lies below the (maxActive + 1)-th largest (as floats) are detached.  Radix select on the float keys, as in k_decode.
         if (a.maxActive > 0 && sh.tail > a.maxActive) {
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
                     c.seq[i] = -1; pos[n] = -1; imax[n] = LZERO; ex[n] = o_null();
                     const int4 ni = N.nodeInfo[n];
                     const int nt = ((ni.x & 15) == HTKAMD_NODE_HMM) ? ((ni.x >> 4) & 255) - 1 : 1;
                     for (int q = 0; q < nt; q++) tok[ni.y + q] = o_null();
                  }
            }
            __syncthreads();
         }
         // ---- the blanks out of the list (its order stays): seq -> the other buffer
         {
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
         // ---- pass 1: StepInst1 on every instance (no order in it: the beams' tops are maxima)
         const float gT = thr[0];                           // threshold of the previous frame
         double myGen = LZERO, myWord = LZERO;
         const int nLive = sh.tail;
         for (int i = tid; i < nLive; i += ORD_THREADS) {
            const int n = c.seq[i];
            const int4 ni = N.nodeInfo[n];
            if ((ni.x & 15) != HTKAMD_NODE_HMM) { tok[ni.y] = o_null(); ex[n] = o_null(); imax[n] = LZERO; continue; }     // StepWord1 (HRec.c:1038)
            const int NS = (ni.x >> 4) & 255, t0 = ni.y;
            const float *tp = tpBase + ni.z;
            Tok s[DEC_MAXN];
            bool live = false;
#pragma unroll
            for (int q = 1; q < DEC_MAXN; q++) { s[q] = o_null(); if (q < NS) s[q] = tok[t0 + q - 1]; }
#pragma unroll
            for (int q = 1; q < DEC_MAXN; q++) if (q < NS && s[q].like > LSMALL) live = true;
            Tok exT = o_null();
            double mx = LZERO;
            if (live) {
               Tok nw[DEC_MAXN];
#pragma unroll
               for (int j = 2; j < DEC_MAXN; j++) {
                  nw[j] = o_null();
                  if (j < NS) {
                     int lo = 1, hi = NS - 1;                 // CreateSEIndex (HRec.c:1403)
                     while (lo < NS && !(tp[(lo - 1) * NS + (j - 1)] > LSMALL)) lo++;
                     while (hi > 1 && !(tp[(hi - 1) * NS + (j - 1)] > LSMALL)) hi--;
                     if (lo > hi) { lo = 1; hi = NS - 1; }
                     Tok best = s[1]; double bl = LZERO;
#pragma unroll
                     for (int q = 1; q < DEC_MAXN; q++)
                        if (q >= lo && q <= hi) {
                           const double cc = s[q].like + tp[(q - 1) * NS + (j - 1)];
                           if (q == lo || cc > bl) { best = s[q]; bl = cc; }
                        }
                     best.like = bl;
                     if (best.like > gT) {
                        const int st = N.hmmState[ni.w + (j - 2)];
                        best.like += a.score[ud.score0 + (size_t)(t - 1) * a.ns + N.stateSlot[st]];
                        nw[j] = best;
                        if (best.like > mx) mx = best.like;
                     }
                  }
               }
               {
                  int lo = 2, hi = NS - 1;
                  while (lo < NS && !(tp[(lo - 1) * NS + (NS - 1)] > LSMALL)) lo++;
                  while (hi > 1 && !(tp[(hi - 1) * NS + (NS - 1)] > LSMALL)) hi--;
                  if (lo > hi) { lo = 2; hi = NS - 1; }
                  Tok best = nw[2]; double bl = LZERO;
#pragma unroll
                  for (int q = 2; q < DEC_MAXN; q++)
                     if (q >= lo && q <= hi) {
                        const double cc = nw[q].like + tp[(q - 1) * NS + (NS - 1)];
                        if (q == lo || cc > bl) { best = nw[q]; bl = cc; }
                     }
                  best.like = bl;
                  if (best.like > LSMALL) {
                     exT = best;
                     const double w = best.like + N.wdlk[n];
                     if (w > myWord) myWord = w;
                  }
               }
               tok[t0] = o_null();                            // entry consumed
#pragma unroll
               for (int j = 2; j < DEC_MAXN; j++) if (j < NS) tok[t0 + j - 1] = nw[j];
               if (mx > myGen) myGen = mx;
            }
            ex[n] = exT; imax[n] = (double)(float)mx;         // inst->max is a LogFloat (HRec.c:138)
         }
         const double genMax = o_block_max(myGen, red);
         const double wordMax = o_block_max(myWord, red2);
         if (tid == 0) {
            float w = (float)(wordMax - a.wordBeam); if (w < (float)LSMALL) w = (float)LSMALL;
            float g = (float)(genMax - a.genBeam); if (g < (float)LSMALL) g = (float)LSMALL;
            thr[0] = g; thr[1] = w;
         }
         __threadfence_block();
         __syncthreads();
      }
      // ---- pass 2 (HRec.c:2011-2021; at t = 0 StartRecognition's): wavefront 0 walks the list while it changes
      if (wv == 0) {
         const float gT = thr[0], wT = thr[1];
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
               if (n < 0) continue;                           // moved or detached since the chunk was fetched
               const float nmax = vs->chunkMax[i];
               const int4 ni = N.nodeInfo[n];
               const int kind = ni.x & 15, NS = (ni.x >> 4) & 255, t0 = ni.y;
               if (nmax < gT) {                               // DetachInst (HRec.c:1270): every token of the instance goes
                  if (lane == 0) { o_blank(c, n); pos[n] = -1; imax[n] = LZERO; ex[n] = o_null(); }
                  const int nt = (kind == HTKAMD_NODE_HMM) ? NS - 1 : 1;
                  if (lane < nt) tok[t0 + lane] = o_null();
                  __threadfence_block();
                  continue;
               }
               // StepInst2 (HRec.c:1360)
               Tok e;
               if (kind == HTKAMD_NODE_WORD) {               // StepWord2 (HRec.c:1046): a Path record per call
                  const Tok st = tok[t0];
                  e = st;
                  e.like += a.wordPen;
                  e.like += N.pronProb[n] * a.prScale;
                  const int pid = vs->nPath;
                  if ((size_t)pid >= pathCap) { if (lane == 0) vs->status = -4; break; }
                  if (lane == 0) {
                     pathPrev[pid] = st.path; pathLike[pid] = e.like; pathLm[pid] = e.lm; pathNode[pid] = n; pathFrame[pid] = t;
                     vs->nPath = pid + 1;
                  }
                  e.path = pid; e.lm = 0.0f;
                  if (lane == 0) ex[n] = e;
               } else if (kind == HTKAMD_NODE_NULL) {
                  e = tok[t0];
                  if (lane == 0) ex[n] = e;
               } else {
                  e = ex[n];
                  if ((ni.x >> 12) & 1) {                    // tee model: StepHMM2 (HRec.c:790)
                     const Tok st = tok[t0];
                     const double cc = st.like + tpBase[ni.z + (NS - 1)];
                     if (cc > e.like) { e = st; e.like = cc; if (lane == 0) ex[n] = e; }
                  }
               }
               Tok tk = e;
               if (kind != HTKAMD_NODE_HMM && tk.like < wT) tk = o_null();
               if (tk.like > gT) {
                  const int k0 = N.linkOff[n], k1 = N.linkOff[n + 1];
                  const bool dup = N.dupDest[n] != 0;
                  for (int kb = k0; kb < k1; kb += 64) {
                     const int k = kb + lane;
                     const bool act = k < k1;
                     const int d = act ? N.linkDest[k] : 0;
                     const float lm = act ? N.linkLike[k] : 0.0f;
                     Tok x = tk;
                     x.like = tk.like + lm * a.lmScale; x.lm = tk.lm + lm;
                     const bool pass = act && x.like > gT;
                     // AttachInst for the destinations without an instance, link by link (appending is what orders the list)
                     unsigned long long need = __ballot(pass && pos[d] < 0);
                     while (need) {
                        const int j = __ffsll((long long)need) - 1;
                        need &= need - 1;
                        const int dj = __shfl(d, j);
                        if (lane == 0 && pos[dj] < 0) o_attach(c, dj);       // (a destination met twice among the links is attached once)
                        __threadfence_block();
                     }
                     if (vs->status != 0) break;
                     // SetEntryState (HRec.c:1303): strict >, the first of equal tokens stays; the instance's max follows
                     unsigned long long todo = __ballot(pass);
                     if (!dup) todo = pass ? (1ull << lane) : 0ull;          // distinct destinations: every lane its own, at once
                     while (todo) {
                        const int j = dup ? __ffsll((long long)todo) - 1 : lane;
                        todo = dup ? (todo & (todo - 1)) : 0ull;
                        if (lane == j) {
                           const int td = N.nodeInfo[d].y;
                           Tok cur = tok[td];
                           if (x.like > cur.like) { tok[td] = x; cur = x; }
                           const double m0 = imax[d];
                           if (cur.like > m0) {
                              const float nm = (float)cur.like;
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

   // ---- CompleteRecognition (HRec.c:2054) + LatFromPaths (:1512) + TranscriptionFromLattice (:2176) for the 1-best chain
   if (tid == 0) {
      int nW = 0;
      // (a capacity of the LIST walk is no reason to lose an answer the batch kernel has already given -- a tie-free order is
      // one valid order of HRec's; the status codes -4 / -5 / -6 then never reach the caller in the default mode)
      if (sh.status != 0 && oa.keepFast && a.nWords[u] >= 0) return;
      a.total[u] = LZERO; a.finalLm[u] = 0.0f;
      if (sh.status != 0) nW = sh.status;
      else {
         const Tok fin = (pos[N.final] >= 0) ? ex[N.final] : o_null();
         const int fp = fin.path;
         if (fp >= 0) {
            a.total[u] = fin.like; a.finalLm[u] = fin.lm;
            for (int p = fp; p >= 0; p = pathPrev[p]) nW++;
            if (nW > a.maxWords) nW = -3;
            else {
               int w = nW;
               for (int p = fp; p >= 0;) {
                  const int prev = pathPrev[p];
                  const double prlk = (prev >= 0) ? pathLike[prev] : 0.0;
                  const double wp = a.wordPen;
                  const float plm = pathLm[p];
                  float aclike = (float)(pathLike[p] - prlk - plm * a.lmScale - wp);
                  const int node = pathNode[p];
                  const float pr = N.pronProb[node];
                  aclike -= pr * a.prScale;
                  const float sc = (float)((double)((aclike * 1.0f + plm * a.lmScale) + pr * a.prScale) + (double)a.wordPen);
                  w--;
                  a.wordPron[ud.out0 + w] = N.model[node];
                  a.wordEnd[ud.out0 + w] = pathFrame[p];
                  a.wordStart[ud.out0 + w] = (prev >= 0) ? pathFrame[prev] : 0;
                  a.wordScore[ud.out0 + w] = sc;
                  a.wordLm[ud.out0 + w] = plm;
                  a.wordAc[ud.out0 + w] = aclike;
                  a.wordLike[ud.out0 + w] = pathLike[p];
                  p = prev;
               }
            }
         } else nW = -1;
      }
      a.nWords[u] = nW;
   }
}

int htkamd_launch_decode_ord(const OrdArgs &a, int nSel, hipStream_t s)
{
   if (nSel <= 0) return HTKAMD_OK;
   hipLaunchKernelGGL(k_decode_ord, dim3(nSel), dim3(ORD_THREADS), 0, s, a);
   hipError_t e = hipGetLastError();
   if (e != hipSuccess) { htkamd_set_error("decode_ord: launch: %s", hipGetErrorString(e)); return HTKAMD_EHIP; }
   return HTKAMD_OK;
}
