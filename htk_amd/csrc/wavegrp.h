// wavegrp.h -- the W wavefronts that work on one utterance as one group of 64*W lanes (fb_wave.hip, viterbi.hip).
// W = 1 compiles to plain wave shuffles and ballots; W > 1 exchanges the few values that cross a 64-lane boundary through LDS with
// one LDS-only barrier per exchange.  Every member that exchanges must be called by all wavefronts of the group, in the same order.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ int lowest_set(unsigned long long m) { return __ffsll((long long)m) - 1; }       // -1 if none
__device__ __forceinline__ int highest_set(unsigned long long m) { return m ? 63 - __clzll((long long)m) : -1; }
// lanes [lo..hi] (0-based, inclusive) as a mask
__device__ __forceinline__ unsigned long long lane_range(int lo, int hi)
{
   if (hi < lo || hi < 0 || lo > 63) return 0ull;
   const unsigned long long upTo = (hi >= 63) ? ~0ull : ((1ull << (hi + 1)) - 1);
   const unsigned long long below = (lo <= 0) ? 0ull : ((1ull << lo) - 1);
   return upTo & ~below;
}

// ---- the W wavefronts of one utterance as one group of 64*W "lanes" (group lane gl = 64*wave + lane holds model gl+1)
template <int W> struct MaskW {                 // one bit per group lane
   unsigned long long w[W];
   // (the word by selects: indexed by a run-time value the words went to scratch memory, a store and a load per step of the pruning beta pass)
   __device__ __forceinline__ bool bit(int i) const
   {
      const int kw = (i >> 6) & (W - 1);
      unsigned long long v = 0;
#pragma unroll
      for (int k = 0; k < W; k++) v |= w[k] & (0ull - (unsigned long long)(kw == k));      // (masks, not selects: a chain of selects is folded back into an indexed load)
      return ((v >> (i & 63)) & 1ull) != 0;
   }
   __device__ __forceinline__ int highest() const
   {
#pragma unroll
      for (int k = W - 1; k >= 0; k--) if (w[k]) return 64 * k + highest_set(w[k]);
      return -1;
   }
   __device__ __forceinline__ int lowest() const
   {
#pragma unroll
      for (int k = 0; k < W; k++) if (w[k]) return 64 * k + lowest_set(w[k]);
      return -1;
   }
   __device__ __forceinline__ MaskW operator&(const MaskW &o) const
   {
      MaskW r;
#pragma unroll
      for (int k = 0; k < W; k++) r.w[k] = w[k] & o.w[k];
      return r;
   }
   __device__ __forceinline__ MaskW operator~() const
   {
      MaskW r;
#pragma unroll
      for (int k = 0; k < W; k++) r.w[k] = ~w[k];
      return r;
   }
   static __device__ __forceinline__ MaskW range(int lo, int hi)          // group lanes [lo..hi], 0-based inclusive
   {
      MaskW r;
#pragma unroll
      for (int k = 0; k < W; k++) r.w[k] = lane_range(lo - 64 * k, hi - 64 * k);
      return r;
   }
};

// Barrier of the utterance's wavefronts for the LDS exchange only.  __syncthreads() would also wait for every outstanding global
// load (s_waitcnt vmcnt(0)), i.e. for the beta column and scores requested two frames ahead at the top of each step, and put the
// HBM latency back into every step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int W> struct Grp {
   unsigned long long *x;      // LDS exchange slots [2][W][4]
   float *a1;                  // LDS: a_1N of every model of the chain, [64*W]
   int wave, lane, ph;
   __device__ __forceinline__ unsigned long long *slot(int w) const { return x + ((ph * W + w) << 2); }
   // value of group lane gl+d (d = 1 or 2); the last lanes of the last wave keep their own
   __device__ __forceinline__ void down2(double v, double &r1, double &r2)
   {
      r1 = __shfl_down(v, 1); r2 = __shfl_down(v, 2);
      if constexpr (W > 1) {
         if (lane < 2) slot(wave)[lane] = (unsigned long long)__double_as_longlong(v);
         lds_barrier();
         if (wave + 1 < W) {
            const unsigned long long *n = slot(wave + 1);
            if (lane == 63) { r1 = __longlong_as_double((long long)n[0]); r2 = __longlong_as_double((long long)n[1]); }
            if (lane == 62) r2 = __longlong_as_double((long long)n[0]);
         }
         ph ^= 1;
      }
   }
   __device__ __forceinline__ void up2(double v, double &r1, double &r2)
   {
      r1 = __shfl_up(v, 1); r2 = __shfl_up(v, 2);
      if constexpr (W > 1) {
         if (lane >= 62) slot(wave)[lane - 62] = (unsigned long long)__double_as_longlong(v);
         lds_barrier();
         if (wave > 0) {
            const unsigned long long *n = slot(wave - 1);
            if (lane == 0) { r1 = __longlong_as_double((long long)n[1]); r2 = __longlong_as_double((long long)n[0]); }
            if (lane == 1) r2 = __longlong_as_double((long long)n[1]);
         }
         ph ^= 1;
      }
   }
   __device__ __forceinline__ double down1(double v) { double r1, r2; down2(v, r1, r2); return r1; }
   __device__ __forceinline__ double up1(double v) { double r1, r2; up2(v, r1, r2); return r1; }
   __device__ __forceinline__ double upBy2(double v) { double r1, r2; up2(v, r1, r2); return r2; }
   __device__ __forceinline__ MaskW<W> ballot(bool p)
   {
      MaskW<W> r;
      const unsigned long long b = __ballot(p);
      if constexpr (W == 1) r.w[0] = b;
      else {
         if (lane == 0) slot(wave)[0] = b;
         lds_barrier();
#pragma unroll
         for (int k = 0; k < W; k++) r.w[k] = slot(k)[0];
         ph ^= 1;
      }
      return r;
   }
   __device__ __forceinline__ double maxall(double g)
   {
      for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_xor(g, o));
      if constexpr (W > 1) {
         if (lane == 0) slot(wave)[0] = (unsigned long long)__double_as_longlong(g);
         lds_barrier();
#pragma unroll
         for (int k = 0; k < W; k++) g = fmax(g, __longlong_as_double((long long)slot(k)[0]));
         ph ^= 1;
      }
      return g;
   }
   __device__ __forceinline__ double bcast(double v, int gl)              // value of group lane gl
   {
      if constexpr (W == 1) return __shfl(v, gl);
      else {
         if (wave == (gl >> 6) && lane == (gl & 63)) slot(0)[0] = (unsigned long long)__double_as_longlong(v);
         lds_barrier();
         const double r = __longlong_as_double((long long)slot(0)[0]);
         ph ^= 1;
         return r;
      }
   }
   // a_1N of model gl+1: a wave shuffle of the lane's own value when there is one wave, the LDS copy otherwise
   __device__ __forceinline__ float a1N_at(float mine, int gl) const { if constexpr (W == 1) return __shfl(mine, gl); else return a1[gl]; }
};

