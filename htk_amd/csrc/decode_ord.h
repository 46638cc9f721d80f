// decode_ord.h -- HRec's instance list on the device: what the exact-order kernels (decode_ord.hip: 1-best; decode_n.hip: k_decode_ord_n,
// token sets) share.  See decode_ord.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include "decode.h"

#define ORD_THREADS 256
#define ORD_STACK 64               /* depth of ReOrderList's recursion (a chain of zero-time nodes) */

__device__ __forceinline__ Tok o_null() { Tok t; t.like = LZERO; t.lm = 0.0f; t.path = -1; return t; }

__device__ __forceinline__ double o_block_max(double v, double *red)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = (w > v) ? w : v; }
   __syncthreads();
   if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
   __syncthreads();
   double r = red[0];
   for (int i = 1; i < ORD_THREADS / 64; i++) r = (red[i] > r) ? red[i] : r;
   return r;
}

// what the walking wavefront shares: LDS, written by lane 0 or by all lanes with the same value
struct OrdShared {
   int tail;                      // entries of seq in use
   int nPath;                     // Path records allocated
   int status;                    // 0, or the error the utterance ends with
   int base, cn;                  // the chunk of seq the walk holds: [base, base + cn)
   int chunkNode[64];
   float chunkMax[64];
   int stkNode[ORD_STACK], stkCur[ORD_STACK];
};

struct OrdCtx {
   const DecNet *N;
   volatile int *seq; volatile int *pos; volatile unsigned char *ooo;
   volatile double *imax;
   int seqCap;
   OrdShared *sh;
};

// MoveToRecent (HRec.c:1123) / the list part of DetachInst: by ONE lane
__device__ __forceinline__ void o_blank(const OrdCtx &c, int n)
{
   const int p = c.pos[n];
   c.seq[p] = -1;
   if (p >= c.sh->base && p < c.sh->base + c.sh->cn) ((volatile int *)c.sh->chunkNode)[p - c.sh->base] = -1;
}
__device__ __forceinline__ bool o_append(const OrdCtx &c, int n)
{
   const int tl = ((volatile OrdShared *)c.sh)->tail;
   if (tl >= c.seqCap) { c.sh->status = -5; return false; }
   c.seq[tl] = n; c.pos[n] = tl; ((volatile OrdShared *)c.sh)->tail = tl + 1;
   c.ooo[n] = 1;
   return true;
}

// ReOrderList (HRec.c:1152) on node n0, whose instance has just been appended: by ONE lane, the recursion on an explicit stack
__device__ void o_reorder(const OrdCtx &c, int n0)
{
   const DecNet &N = *c.N;
   OrdShared *sh = c.sh;
   int sp = 0;
   sh->stkNode[0] = n0; sh->stkCur[0] = -1; sp = 1;
   while (sp > 0) {
      const int n = sh->stkNode[sp - 1];
      const int cur = sh->stkCur[sp - 1];
      const int k0 = N.linkOff[n], nt = N.nTr0[n];
      if (cur < 0) {
         if (c.pos[n] < 0 || !c.ooo[n]) { sp--; continue; }
         c.ooo[n] = 0;
         for (int k = 0; k < nt; k++) {
            const int d = N.linkDest[k0 + k];
            if (c.pos[d] >= 0) { o_blank(c, d); if (!o_append(c, d)) return; }
         }
         sh->stkCur[sp - 1] = 0;
      } else if (cur >= nt) sp--;
      else {
         sh->stkCur[sp - 1] = cur + 1;
         const int d = N.linkDest[k0 + cur];
         if (c.pos[d] >= 0) {
            if (sp >= ORD_STACK) { sh->status = -6; return; }
            sh->stkNode[sp] = d; sh->stkCur[sp] = -1; sp++;
         }
      }
   }
}

// AttachInst (HRec.c:1200): by ONE lane.  The node's tokens are null already (DetachInst and the start leave them so).
__device__ __forceinline__ void o_attach(const OrdCtx &c, int n)
{
   if (!o_append(c, n)) return;
   if (c.N->nTr0[n] > 0) o_reorder(c, n);
   else c.ooo[n] = 0;
}

