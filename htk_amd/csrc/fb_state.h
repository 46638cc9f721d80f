// fb_state.h -- what the lane-per-chain-state kernels share (fb_state.hip: log domain; fb_lin.hip: scaled linear domain):
// the per-lane description of a chain state, the exchange arrays' padding, the staged score rows and the beta layout macros.
#pragma once
#include <hip/hip_runtime.h>
#include "internal.h"
#include "kernels.h"
#include "wavegrp.h"

#define EXPFLOOR (-100.0)
#define SPAD 8                        // padding lanes on both sides of the exchange arrays (offsets reach -5..+5)

typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));

// ---- one exchange array: every lane writes its own slot, then reads slots of other lanes
template <int W> struct Xc {
   double *b;                          // LDS [64*W + 2*SPAD]
   int gl;
   __device__ __forceinline__ void put(double v) { b[SPAD + gl] = v; }
   __device__ __forceinline__ double at(int d) const { return b[SPAD + gl + d]; }
   __device__ __forceinline__ double lane(int l) const { return b[SPAD + l]; }
};
template <int W> __device__ __forceinline__ void xsync()
{
   if constexpr (W > 1) lds_barrier();
   else asm volatile("" ::: "memory");            // one wavefront: LDS operations execute in order
}

// per-lane description of a chain state
struct StateRegs {
   int q, j, N, mi;                    // model (1-based), state in the model (2..N-1), states of the model, index into the per-model tables
   float aOut[5], aIn[5];              // a_{j,j+d} and a_{j+d,j} for d = -2..2 (log-zero outside the model)
   float aExit, aEntry;                // a_{jN}, a_{1j}
   float aEntryOf[3], aExitOf[3];      // first lane: a_{1,2+k}; last lane: a_{N-1-k,N}
   bool first, last;
};

__device__ __forceinline__ void load_state(StateRegs &s, const FbArgs &a, const UttDesc &ud, int gl, bool valid)
{
   s.q = 0; s.j = 2; s.N = 3; s.mi = 0; s.first = false; s.last = false;
   s.aExit = (float)LZERO; s.aEntry = (float)LZERO;
#pragma unroll
   for (int d = 0; d < 5; d++) { s.aOut[d] = (float)LZERO; s.aIn[d] = (float)LZERO; }
#pragma unroll
   for (int k = 0; k < 3; k++) { s.aEntryOf[k] = (float)LZERO; s.aExitOf[k] = (float)LZERO; }
   if (!valid) return;
   s.q = a.sQ[ud.slot0 + gl];
   s.mi = ud.q0 + s.q - 1;
   s.N = a.mN[s.mi];
   s.j = gl - a.mSlot0[s.mi] + 2;
   const float *tp = a.transP + a.mTp[s.mi];
   const int N = s.N, j = s.j;
   s.first = j == 2; s.last = j == N - 1;
#pragma unroll
   for (int d = -2; d <= 2; d++) {
      const int o = j + d;
      if (o >= 2 && o <= N - 1) { s.aOut[d + 2] = tp[(j - 1) * N + (o - 1)]; s.aIn[d + 2] = tp[(o - 1) * N + (j - 1)]; }
   }
   s.aExit = tp[(j - 1) * N + (N - 1)];
   s.aEntry = tp[j - 1];
#pragma unroll
   for (int k = 0; k < 3; k++) {
      if (s.first && 2 + k <= N - 1) s.aEntryOf[k] = tp[2 + k - 1];
      if (s.last && N - 1 - k >= 2) s.aExitOf[k] = tp[(N - 1 - k - 1) * N + (N - 1)];
   }
}

// scores of this lane's state: 4 frames per 16-byte load, one block requested ahead, parked in a wave-private LDS slot
struct ObsRow {
   float *lds;                 // this wave's [2][64][4]
   const float *row;
   f4s R;
   int lane;
   __device__ __forceinline__ void load(int blk) { if (row) R = *(const f4s *)(row + 4 * blk); }
   __device__ __forceinline__ void park(int blk) { *(f4s *)(lds + ((((blk & 1) * 64) + lane) << 2)) = R; }
   __device__ __forceinline__ float get(int f) const { return row ? lds[(((((f >> 2) & 1) * 64) + lane) << 2) + (f & 3)] : 0.0f; }
   // the same for callers that give EVERY lane a row (lanes past the chain: any row of the block): no lane-divergent branch
   __device__ __forceinline__ void load_all(int blk) { R = *(const f4s *)(row + 4 * blk); }
   __device__ __forceinline__ float get_all(int f) const { return lds[(((((f >> 2) & 1) * 64) + lane) << 2) + (f & 3)]; }
};

#define BETA_S(t) (a.betaW[ud.betaW0 + (size_t)((t) - 1) * L + gl])
#define BETA_E(t) (a.betaW[ud.betaW0 + (size_t)T * L + (size_t)((t) - 1) * L + gl])

// Transitions of the models on either side of this lane's model, read straight from the batch tables: a lane derives its model's
// entry value (alpha: the exit value of the model before it) and exit value (beta: the entry value of the model after it) from the
// neighbour model's published state values itself, so a step needs ONE exchange instead of a second round trip through the lanes
// that own those values.
__device__ __forceinline__ void load_neighbours(float aExitPrev[3], float aEntryNext[3], const FbArgs &a, const UttDesc &ud, const StateRegs &s, bool valid)
{
#pragma unroll
   for (int k = 0; k < 3; k++) { aExitPrev[k] = (float)LZERO; aEntryNext[k] = (float)LZERO; }
   if (!valid) return;
   if (s.q > 1) {
      const int mi = s.mi - 1, N = a.mN[mi];
      const float *tp = a.transP + a.mTp[mi];
#pragma unroll
      for (int k = 0; k < 3; k++) if (N - 1 - k >= 2) aExitPrev[k] = tp[(N - 1 - k - 1) * N + (N - 1)];
   }
   if (s.q < ud.Q) {
      const int mi = s.mi + 1, N = a.mN[mi];
      const float *tp = a.transP + a.mTp[mi];
#pragma unroll
      for (int k = 0; k < 3; k++) if (2 + k <= N - 1) aEntryNext[k] = tp[2 + k - 1];
   }
}

