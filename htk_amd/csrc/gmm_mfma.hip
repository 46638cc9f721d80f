// gmm_mfma.hip -- K1m: GMM state log-likelihoods with the Mahalanobis contraction on the matrix cores.
//
// Same quantity as gmm_exact.hip (IDOutP HModel.c:5420 + ShStrP mixture loop HFB.c:949-960), computed in the
// expanded form
//     log( w_m N(x; mu_m, var_m) ) = cinit_m + sum_i ( -0.5*ivar_mi * x_i^2  +  mu_mi*ivar_mi * x_i )
//     cinit_m = log w_m - 0.5*( gConst_m + sum_i mu_mi^2 * ivar_mi )
// (all coefficients pre-multiplied by log2(e) on the host, so that the mixture sum runs on v_exp_f32 / v_log_f32 directly)
// so that frames x Gaussians is a GEMM  [x^2 | x] (T x 2D)  *  W (2D x M)  accumulated on top of cinit, followed by
// a float log-sum-exp over the M columns of a state.  This is the TOLERANCE path (HERest: alpha/beta and the
// re-estimated parameters to 1e-4 relative, tests/test_gpu_parity.py); scores differ from the reference's float
// sum by ~1e-4 absolute (the reference's own float rounding noise is of that size), so the Viterbi path -- whose
// bar is a bit-exact alignment -- and the default of htkamd_fb_execute stay on gmm_exact.hip.
//
// MI355X mapping.  v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate).  A workgroup of four waves owns one task
// (128 frames x up to 16 chain states); each wave owns 32 of the frames (two column tiles of 16) and all four walk
// the task's states together.  The GAUSSIANS are the rows of the product and the FRAMES its columns:
//   A[row = lane&15][k = lane>>4] : the state's 16 components -- host-built fragment table [tile][step][lane]
//       (model.hip mfma_refresh).  One copy per workgroup, staged global -> registers -> LDS a tile ahead of use
//       (double buffer, one barrier per tile), read by all four waves: parameter traffic per 128 frames, not per wave.
//   B[k = lane>>4][col = lane&15] : the wave's frames.  K index 4*s+kq carries dimension 2*s+(kq>>1), as x^2 for
//       even kq and x for odd kq: ceil(D/2) registers per column tile, loaded once per task (2*20 VGPRs at D=39).
//   C[row = 4*(lane>>4)+r][col = lane&15] : a lane ends with 4 components of one frame; accumulators start at cinit
//       (table rows NS..NS+3), so the GEMM result IS log(w N).
// The mixture log-sum-exp is then 4 values in-lane + two row swaps (v_permlane16_swap / v_permlane32_swap) for max and for sum.
// Lanes 0..31 store frames fw..fw+31 of the state's output row (128 contiguous bytes).  A short last piece leaves
// whole waves without frames; they skip the arithmetic and only take part in the staging.
// States with more than 16 components take several tiles, merged with a running (max, sum).
//
// Measured (tools/ubench/mfma_rate.hip): the fp32 MFMA sustains 146-150 TFLOP/s (34 cycles per instruction and SIMD),
// and VALU instructions issued beside it do NOT overlap -- each costs ~3.3 more cycles per SIMD even from other
// waves (fp32 MFMA runs at the FP32 vector rate) -- so every non-MFMA instruction in the state loop is pure cost.
//
// Roofline: 2*16*(4*NS) flop per (frame, tile) on MFMA = 2560 flop at D=39 against the algorithmic
// M*(4*D+8) = 2624; peak 157.3 TFLOP/s (fp32 matrix).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) int cint;

#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define LOG2(x) __builtin_amdgcn_logf(x)
#define MFMA_COL_TILES 2          /* 16-frame column tiles per wave: 4 waves x 32 frames = one 128-frame task */

// Combining a value across the four 16-lane rows of the wave with gfx950's row-swap instructions (VALU, no LDS round trip):
// v_permlane16_swap(v, v) leaves {rows 0,0,2,2} and {rows 1,1,3,3}; v_permlane32_swap(v, v) {lower, lower} and {upper, upper}:
// op(first, second) is op(lane, lane ^ 16) resp. op(lane, lane ^ 32) in every lane.
__device__ __forceinline__ float rows_max(float v)
{
   auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
   auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v)
{
   auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
   auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int NS>
__global__ __launch_bounds__(256, 4) void k_score_mfma(ScoreArgs a)
{
   constexpr int TW = (NS + 8) * 64;                  // floats per fragment tile: NS K-steps, the accumulators' start, the closing constant
   constexpr int PT = (TW + 255) / 256;               // floats staged per thread
   __shared__ float wbuf[2][TW];
   __shared__ int taskSh;
   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const int col = lane & 15, kq = lane >> 4;
   const int D = a.D;
   cint *slotState = (cint *)a.slotState;
   cint *stateTileOff = (cint *)a.stateTileOff;

   for (;;) {
      if (tid == 0) taskSh = atomicAdd(a.taskCounter, 1);
      __syncthreads();
      const int task = __builtin_amdgcn_readfirstlane(taskSh);
      if (task >= a.nTasks) break;
      const ScoreTask tk = a.tasks[task];
      const int fw = 32 * wv;                         // this wave's first frame in the tile
      const bool active = fw < tk.nFrames;            // a short last piece leaves whole waves without frames: they only stage

      // first fragment tile -> LDS (synchronised by the barrier below)
      int tile = stateTileOff[slotState[tk.slot0]];
      {
         const float *W = a.mfmaTab + (size_t)tile * TW;
#pragma unroll
         for (int j = 0; j < PT; j++)
            if (j * 256 + tid < TW) wbuf[0][j * 256 + tid] = W[j * 256 + tid];
      }

      // B operand: this lane's frame (col) of each column tile, K slice kq
      float xf[MFMA_COL_TILES][NS];
      if (active)
#pragma unroll
      for (int ft = 0; ft < MFMA_COL_TILES; ft++) {
         int f = fw + ft * 16 + col;
         if (f > tk.nFrames - 1) f = tk.nFrames - 1;
         const float *row = a.X + (size_t)(tk.frame0 + f) * D;
#pragma unroll
         for (int s = 0; s < NS; s++) {
            int dim = 2 * s + (kq >> 1);
            const bool pad = dim >= D;
            if (pad) dim = D - 1;
            float v = row[dim];
            if (pad) v = 0.0f;
            xf[ft][s] = (kq & 1) ? v : v * v;
         }
      }
      __syncthreads();

      int buf = 0;
      for (int k = 0; k < tk.nSlots; k++) {
         const int st = slotState[tk.slot0 + k];
         const int t1 = stateTileOff[st + 1];
         const int nextFirst = (k + 1 < tk.nSlots) ? stateTileOff[slotState[tk.slot0 + k + 1]] : -1;
         float rM[MFMA_COL_TILES], rS[MFMA_COL_TILES];
         bool first = true;
         for (;;) {
            // stage the next tile (of this state, or the first of the next state) while this one is computed
            const int nextTile = (tile + 1 < t1) ? tile + 1 : nextFirst;
            float stg[PT];
            if (nextTile >= 0) {
               const float *W = a.mfmaTab + (size_t)nextTile * TW;
#pragma unroll
               for (int j = 0; j < PT; j++)
                  if (j * 256 + tid < TW) stg[j] = W[j * 256 + tid];
            }
            if (active) {
            float w[NS + 8];
#pragma unroll
            for (int s = 0; s < NS + 8; s++) w[s] = wbuf[buf][s * 64 + lane];
            f4 Cx[MFMA_COL_TILES];
#pragma unroll
            for (int ft = 0; ft < MFMA_COL_TILES; ft++) Cx[ft] = (f4){w[NS], w[NS + 1], w[NS + 2], w[NS + 3]};
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
               for (int ft = 0; ft < MFMA_COL_TILES; ft++)
                  Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], xf[ft][s], Cx[ft], 0, 0, 0);
            }
            // log-sum-exp over the tile's 16 rows: 4 in this lane, the rest in lanes ^16, ^32, ^48
#pragma unroll
            for (int ft = 0; ft < MFMA_COL_TILES; ft++) {
               // the matrix unit rounds the accumulator at its own magnitude after every instruction: it starts at HALF of the expanded
               // form's constant -0.5 sum mu^2 ivar (~ -116 at D = 39) and passes through zero on its way up; the other half and
               // log w - 0.5 gConst are added here, once (starting at the whole constant, ~ -290, cost a factor 3 in accuracy)
               const f4 y = Cx[ft] + (f4){w[NS + 4], w[NS + 5], w[NS + 6], w[NS + 7]};
               // the table is scaled by log2(e): y is a base-2 logarithm, so v_exp_f32 / v_log_f32 apply without a multiply
               float mx = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
               mx = rows_max(mx);
               float sm = (EXP2(y[0] - mx) + EXP2(y[1] - mx)) + (EXP2(y[2] - mx) + EXP2(y[3] - mx));
               sm = rows_sum(sm);
               if (first) { rM[ft] = mx; rS[ft] = sm; }
               else {
                  const float M2 = fmaxf(rM[ft], mx);
                  rS[ft] = rS[ft] * EXP2(rM[ft] - M2) + sm * EXP2(mx - M2);
                  rM[ft] = M2;
               }
            }
            first = false;
            }
            if (nextTile >= 0) {
#pragma unroll
               for (int j = 0; j < PT; j++)
                  if (j * 256 + tid < TW) wbuf[buf ^ 1][j * 256 + tid] = stg[j];
            }
            __syncthreads();
            buf ^= 1;
            tile++;
            if (tile >= t1) break;
         }
         tile = nextFirst;
         // lanes 0..31 (kq = column tile) store frames fw + lane: 128 contiguous bytes
         const float r0 = (rM[0] + LOG2(rS[0])) * 0.69314718055994531f, r1 = (rM[1] + LOG2(rS[1])) * 0.69314718055994531f;
         const float res = (kq == 1) ? r1 : r0;
         float *o = a.out + tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + fw;
         if (active && lane < 32 && fw + lane < tk.nFrames) o[lane] = res;
      }
   }
}

int htkamd_launch_score_mfma(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop)
{
   if (a.nTasks <= 0) return HTKAMD_OK;
   if (!m->d_mfmaTab) { htkamd_set_error("score_mfma: vector size %d not supported by the MFMA path (up to 40)", m->D); return HTKAMD_EMODEL; }
   if (m->mfmaStale) {                             // parameters were re-estimated on the device since the table was built
      int rcr = htkamd_model_refresh_mfma_device(const_cast<htkamd_model *>(m), stream);
      if (rcr) return rcr;
   }
   HIPCHECK(hipMemsetAsync(a.taskCounter, 0, sizeof(int), stream));
   int blocks = a.nTasks;
   if (blocks > 256 * 4) blocks = 256 * 4;      // persistent blocks (4 per CU at <= 128 VGPRs), one task (128 frames x 16 states) at a time
   dim3 grid(blocks), block(256);
   switch (m->mfmaNS) {
   case 20: hipExtLaunchKernelGGL((k_score_mfma<20>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 13: hipExtLaunchKernelGGL((k_score_mfma<13>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 7: hipExtLaunchKernelGGL((k_score_mfma<7>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   default: htkamd_set_error("score_mfma: no kernel for %d K-steps", m->mfmaNS); return HTKAMD_EMODEL;
   }
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}
