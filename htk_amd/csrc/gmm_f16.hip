// gmm_f16.hip -- K1h: GMM state log-likelihoods on the FP16 matrix pipe with two-way operand splitting.
//
// Same quantity, the same expanded form and the same task / tile / log-sum-exp structure as gmm_bf16.hip (K1b):
//     log2( w_m N(x; mu_m, var_m) ) = cinit_m + sum_k a_mk z_k ,   z = (x_0^2, x_0, x_1^2, x_1, ...)
// K1b splits both operands into THREE bf16 pieces (8 significant bits each) and needs six piece products per K chunk for fp32 accuracy.
// An fp16 carries 11 significant bits, so TWO pieces reach fp32's 24 (round to nearest: |a - a1| <= 2^-11 |a|, and the second piece
// takes 11 bits of the remainder: |a - a1 - a2| <= 2^-22 |a|, 2^-24.6 rms -- fp32's own rounding is 2^-24) and THREE products do:
//     a z  =  a1 z1 + (a1 z2 + a2 z1)  +  O(2^-22 |a z|)
// on v_mfma_f32_16x16x32_f16, which runs at the bf16 instruction's rate: half of K1b's matrix instructions, two thirds of its operand
// registers (4 waves per SIMD instead of 3), LDS reads and table bytes.  What fp16 lacks is RANGE (6e-5 .. 65504), so
//   * every k carries a power-of-two scale s_k: the table holds a_k s_k, the frames' side z_k / s_k (exact).  s_k puts the largest
//     |a_k| over the Gaussians and the largest |z_k| the model lets expect (|x| <= |mu| + 8 sigma) at the same height, the square root
//     of their product: (k_f16_range, k_f16_scale, on the device with every table refresh);
//   * the second pieces are carried at 2^11 times their value, a2' = (a - a1) 2^11 -- as large as a1, never subnormal where a1 is not --
//     and their products go to an accumulator of their own, added as Cc 2^-11 at the end.  The leading products a1 z1 (complete
//     squares per chunk, as in K1b) have the main accumulator to themselves: three roundings at its magnitude instead of eighteen;
//   * a scaled value beyond 65504 -- a feature far outside anything the model describes, or a model whose coefficients span more than
//     the format -- raises a flag (ScoreArgs::rangeFlag) instead of a silent infinity: htkamd_fb_results / htkamd_outp_block_mode
//     return HTKAMD_ERANGE and the caller repeats the pass with HTKAMD_SCORE_BF16, whose pieces have fp32's exponent.
// Measured (bench workload, tools/ubench/score_cmp.py): |score - exact| rms 3.4e-5, max 2.1e-4 against K1b's 3.2e-5 / 1.8e-4 (the final rounding
// to float at |score| ~ 100..500 dominates both); 0.85 ms against K1b's 1.24 ms.
//
// Layout.  K chunks as K1b (NC = ceil(D/15) chunks of 32: 15 dimensions as (x^2, x) pairs, the chunk's -0.5 sum mu^2 ivar at k = 30
// against a constant).  Table per tile of 16 components: [piece 2][chunk NC][lane 64][8 f16], then (log w - 0.5 gConst) log2(e) as
// [lane 64][4 f32] in the accumulator's layout.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) int cint;

#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define LOG2(x) __builtin_amdgcn_logf(x)
#ifndef F16_COL_TILES
#define F16_COL_TILES 2                                 // 16-frame column tiles per wavefront: 2 (four wavefronts per 128-frame task) or 4 (two)
#endif

__device__ __forceinline__ float rows_max_b(float v)
{
   auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
   auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum_b(float v)
{
   auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
   auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
   return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// x = p1 + 2^-11 p2 + O(2^-22 |x|) (rms 2^-24.6), both pieces fp16 (round to nearest even), the difference exact in fp32
#define F16_CORR 2048.0f
__device__ __forceinline__ void split2(float x, _Float16 &p1, _Float16 &p2)
{
   p1 = (_Float16)x;
   p2 = (_Float16)((x - (float)p1) * F16_CORR);
}
__device__ __forceinline__ unsigned int pack2(_Float16 a, _Float16 b)
{
   h2 v; v[0] = a; v[1] = b;
   return __builtin_bit_cast(unsigned int, v);
}

// the model's control block (htkamd_model::d_f16Ctl): float scale[96] of every k (table side), float inverse[96] (frames' side),
// unsigned range[192] (bits of max |a_k|, then of the expected max |z_k|), int flag (HTKAMD_F16_* bits of the last table build), the
// model's sticky flag, and at F16_CTL_MQ a row of k_f16_range's
#define F16_CTL_RANGE 192
#define F16_CTL_FLAG  384
#define F16_CTL_STICKY 511     /* the model's sticky range flag */
#define F16_CTL_MQ    400      /* float bits [48]: per dimension the largest 0.5 mu^2 ivar log2(e) (k_f16_range -> k_f16_scale) */
#define F16_MAX 65504.0f

#ifndef F16_EU
#define F16_EU (F16_COL_TILES > 2 ? 2 : 4)
#endif
#ifndef F16_WPB
#define F16_WPB (8 / F16_COL_TILES)                     // wavefronts per workgroup: together a whole 128-frame task (F16_WPB smaller: the task in parts)
#endif
template <int NC>
__global__ __launch_bounds__(64 * F16_WPB, F16_EU) void k_score_f16(ScoreArgs a)      // second figure: wavefronts per SIMD the register budget is cut for
{
   constexpr int NT = 64 * F16_WPB, FPW = 16 * F16_COL_TILES, HALVES = B16_TASK_FRAMES / (FPW * F16_WPB);
   constexpr int TWB = 2 * NC * 64 * 16 + 64 * 16;     // bytes per fragment tile
   constexpr int TW4 = TWB / 16;                       // 16-byte words per tile
   constexpr int PT = (TW4 + NT - 1) / NT;             // words staged per thread
   __shared__ u4 wbuf[2][TW4];
   __shared__ int taskSh;
   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const int col = lane & 15, kg = lane >> 4;
   const int D = a.D;
   const int dpc = (D + NC - 1) / NC;                  // dimensions per K chunk (<= 15)
   const u4 *tab = (const u4 *)a.f16Tab;
   if (blockIdx.x == 0 && tid == 0 && a.f16Ctl[F16_CTL_FLAG]) atomicOr(a.rangeFlag, a.f16Ctl[F16_CTL_FLAG]);      // what the table build found

   for (;;) {
      if (tid == 0) taskSh = atomicAdd(a.taskCounter, 1);
      __syncthreads();
      const int vtask = __builtin_amdgcn_readfirstlane(taskSh);
      const int task = vtask / HALVES;
      if (task >= a.nTasks) break;
      const ScoreTask tk = a.tasks[task];
      const int fw = FPW * wv + (B16_TASK_FRAMES / HALVES) * (vtask % HALVES);      // this wave's first frame in the tile
      const bool active = fw < tk.nFrames;

      // first tile and tile count of every state of the task, one per lane (tasks hold at most 64 states): the tile loop below reads them
      // with v_readlane instead of chains of dependent scalar loads (two per state, each a round trip to the scalar cache or L2)
      int tFirstV = 0, tEndV = 0;
      if (lane < tk.nSlots) { const int stl = a.slotState[tk.slot0 + lane]; tFirstV = a.stateTileOff[stl]; tEndV = a.stateTileOff[stl + 1]; }
      int tile = __builtin_amdgcn_readlane(tFirstV, 0);
      {
         const u4 *W = tab + (size_t)tile * TW4;
#pragma unroll
         for (int j = 0; j < PT; j++)
            if (j * NT + tid < TW4) wbuf[0][j * NT + tid] = W[j * NT + tid];
      }

      // B operand: this lane's frame (col) of each column tile, the 8 k of lane group kg in every chunk, scaled, in two fp16 pieces
      h8 zb[F16_COL_TILES][2][NC];
      if (active) {
         const float *zs = (const float *)a.f16Ctl + 96;      // 1 / scale of every k
         bool over = false;
         int kgL = kg, colL = col;
         asm volatile("" : "+v"(kgL), "+v"(colL));      // (the indices below are cheap to recompute per task: hoisted out of the task loop they are spilled)
#pragma unroll
      for (int c = 0; c < NC; c++) {
         float is[8];
#pragma unroll
         for (int j = 0; j < 8; j++) is[j] = zs[c * 32 + 8 * kgL + j];
#pragma unroll
      for (int ft = 0; ft < F16_COL_TILES; ft++) {
         int f = fw + ft * 16 + colL;
         if (f > tk.nFrames - 1) f = tk.nFrames - 1;
         const float *row = a.X + (size_t)(tk.frame0 + f) * D;
            const int d0 = dpc * c + 4 * kgL;          // dimensions d0..d0+3 -> k = 32c + 8kg + (0..7) = (x^2, x) pairs
            _Float16 p[2][8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
               int dim = d0 + i;
               const bool pad = 4 * kgL + i >= dpc || dim >= D;
               if (pad) dim = D - 1;
               float v = row[dim];
               if (pad) v = 0.0f;
               float v2 = v * v;
               if (i == 3 && kgL == 3) v2 = 1.0f;       // k = 30: the constant that meets the chunk's -0.5 sum mu^2 ivar
               v2 *= is[2 * i]; v *= is[2 * i + 1];
               over = over || !(v2 <= F16_MAX) || !(fabsf(v) <= F16_MAX);
               split2(v2, p[0][2 * i], p[1][2 * i]);
               split2(v, p[0][2 * i + 1], p[1][2 * i + 1]);
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
               u4 w;
               w[0] = pack2(p[s][0], p[s][1]); w[1] = pack2(p[s][2], p[s][3]);
               w[2] = pack2(p[s][4], p[s][5]); w[3] = pack2(p[s][6], p[s][7]);
               zb[ft][s][c] = __builtin_bit_cast(h8, w);
            }
         }
      }
         if (over) atomicOr(a.rangeFlag, HTKAMD_F16_EFEAT);
      }
      __syncthreads();

      int buf = 0;
      for (int k = 0; k < tk.nSlots; k++) {
         const int t1 = __builtin_amdgcn_readlane(tEndV, k);
         const int nextFirst = (k + 1 < tk.nSlots) ? __builtin_amdgcn_readlane(tFirstV, (k + 1) & 63) : -1;
         float rM[F16_COL_TILES], rS[F16_COL_TILES];
         bool first = true;
         for (;;) {
            const int nextTile = (tile + 1 < t1) ? tile + 1 : nextFirst;
            u4 stg[PT];
            if (nextTile >= 0) {
               const u4 *W = tab + (size_t)nextTile * TW4;
#pragma unroll
               for (int j = 0; j < PT; j++)
                  if (j * NT + tid < TW4) stg[j] = W[j * NT + tid];
            }
            if (active) {
               h8 wa[2][NC];
#pragma unroll
               for (int s = 0; s < 2; s++)
#pragma unroll
                  for (int c = 0; c < NC; c++) wa[s][c] = __builtin_bit_cast(h8, wbuf[buf][(s * NC + c) * 64 + lane]);
               const f4 ci = __builtin_bit_cast(f4, wbuf[buf][2 * NC * 64 + lane]);
               f4 Cx[F16_COL_TILES], Cc[F16_COL_TILES];
#pragma unroll
               for (int ft = 0; ft < F16_COL_TILES; ft++) { Cx[ft] = (f4)(0.0f); Cc[ft] = (f4)(0.0f); }
               // corrections (a2 z1 + a1 z2, both carried at 2^11 times their value) in an accumulator of their own, the leading products
               // a1 z1 -- complete squares per chunk -- in the other
#pragma unroll
               for (int c = 0; c < NC; c++)
#pragma unroll
                  for (int ft = 0; ft < F16_COL_TILES; ft++) {
                     Cc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1][c], zb[ft][0][c], Cc[ft], 0, 0, 0);
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][c], zb[ft][0][c], Cx[ft], 0, 0, 0);
                     Cc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][c], zb[ft][1][c], Cc[ft], 0, 0, 0);
                  }
               // log-sum-exp over the tile's 16 rows: 4 in this lane, the rest in lanes ^16, ^32, ^48 (base-2 logs, as K1m)
#pragma unroll
               for (int ft = 0; ft < F16_COL_TILES; ft++) {
                  const f4 y = (Cx[ft] + Cc[ft] * (1.0f / F16_CORR)) + ci;
                  float mx = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
                  mx = rows_max_b(mx);
                  float sm = (EXP2(y[0] - mx) + EXP2(y[1] - mx)) + (EXP2(y[2] - mx) + EXP2(y[3] - mx));
                  sm = rows_sum_b(sm);
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
                  if (j * NT + tid < TW4) wbuf[buf ^ 1][j * NT + tid] = stg[j];
            }
            __syncthreads();
            buf ^= 1;
            tile++;
            if (tile >= t1) break;
         }
         tile = nextFirst;
         float res = 0.0f;
#pragma unroll
         for (int ft = 0; ft < F16_COL_TILES; ft++) {
            const float r = (rM[ft] + LOG2(rS[ft])) * 0.69314718055994531f;
            if (kg == ft) res = r;
         }
         float *o = a.out + tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + fw;
         if (active && lane < FPW && fw + lane < tk.nFrames) o[lane] = res;
      }
   }
}

// ------------------------------------------------------------------------------------ the same on 32 x 32 blocks, states in pairs
// For sets whose states all fit one tile (<= 16 components: htkamd_model::f16Wide).  v_mfma_f32_32x32x16_f16 leaves a lane the rows
// 8b + 4(lane >> 5) + r (b, r = 0..3) of column lane & 31: with the components of state A dealt to the rows whose bit 2 is clear and
// those of state B to the others, lanes 0..31 hold ALL 16 components of state A for their frame and lanes 32..63 those of state B.
// The mixture's log-sum-exp then needs no exchange between lanes (the 16 x 16 form pays four row swaps per column tile), its maximum
// and sum are trees of packed operations within the lane, and every lane ends with one result to store: ~60 vector instructions per
// pair of states where the 16 x 16 form issues ~200, for the same 18 x 32 matrix cycles.
//   Table per tile: [k-step 2 NC][piece 2][k-half 2][component 16][8 f16], then (log w - 0.5 gConst) log2(e) [16 f32]; a k-step is 16 k,
//   chunk c = k-steps 2c, 2c + 1.  In LDS a pair is [k-step][piece][lane 64][8 f16] with lane = 32 k-half + 8 (comp >> 2) + 4 h + (comp & 3),
//   then the two states' constants.
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#ifdef F16W_STAMP            /* experiment builds only (tools/ubench/score_exp.py): cycles per phase and wavefront */
__device__ unsigned long long g_dbg[4096 * 16];
#define STAMP_DECL unsigned long long t_ = __builtin_amdgcn_s_memtime(), t0_ = t_, acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_[i] += n_ - t_; t_ = n_; } while (0)
#define STAMP_COUNT(i, n) acc_[i] += (n)
#define STAMP_FLUSH do { if (lane == 0) { const int w_ = (blockIdx.x * 4 + wv) & 4095; acc_[6] = __builtin_amdgcn_s_memtime() - t0_; for (int i_ = 0; i_ < 10; i_++) g_dbg[w_ * 16 + i_] += acc_[i_]; } } while (0)
extern "C" void htkamd_dbg_zero(void) { void *p; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_dbg)); (void)hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 16); }
extern "C" void htkamd_dbg_read(void *dst) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 4096 * 16); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_COUNT(i, n)
#define STAMP_FLUSH
#endif
#ifndef F16W_PIPE
#define F16W_PIPE 1
#endif
#ifndef F16W_EU
#define F16W_EU 3
#endif
template <int NC>
__global__ __launch_bounds__(256, F16W_EU) void k_score_f16w(ScoreArgs a)
{
   static_assert(B16_TASK_FRAMES == 128, "four wavefronts x 32 frames");
   constexpr int KS = 2 * NC;                          // k-steps of 16
   constexpr int TW4 = KS * 2 * 32 + 4;                // 16-byte words per tile in the table
   constexpr int PW4 = KS * 2 * 64 + 8;                // ... per pair in LDS
   constexpr int PT = (TW4 + 127) / 128;               // words staged per thread (a half workgroup per tile)
   __shared__ u4 wbuf[2][PW4];
   __shared__ float xbuf[128 * 15 * NC];               // the task's 128 feature rows (D <= 15 NC), as they lie in memory
   __shared__ float zsSh[32 * NC];                     // 1 / scale of every k
   __shared__ int taskSh;
   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const int fcol = lane & 31, kh = lane >> 5;
   const int D = a.D;
   const int dpc = (D + NC - 1) / NC;                  // dimensions per K chunk (<= 15)
   const u4 *tab = (const u4 *)a.f16Tab;
   if (blockIdx.x == 0 && tid == 0 && a.f16Ctl[F16_CTL_FLAG]) atomicOr(a.rangeFlag, a.f16Ctl[F16_CTL_FLAG]);      // what the table build found
   if (tid < 32 * NC) zsSh[tid] = ((const float *)a.f16Ctl)[96 + tid];
   // staging: wavefronts 0, 1 bring the first state's tile, 2, 3 the second's; word w of a tile goes to the lane that multiplies it
   const int hsel = __builtin_amdgcn_readfirstlane(wv >> 1), t7 = tid & 127;
   int dst[PT];
#pragma unroll
   for (int j = 0; j < PT; j++) {
      const int w = t7 + 128 * j;
      const int comp = w & 15;
      dst[j] = (w < KS * 64) ? (w >> 5) * 64 + ((w >> 4) & 1) * 32 + 8 * (comp >> 2) + 4 * hsel + (comp & 3) : KS * 128 + hsel * 4 + (w - KS * 64);
   }
   const int fw = 32 * wv;                             // this wave's first frame in the task's tile
   STAMP_DECL;

   for (;;) {
      if (tid == 0) taskSh = atomicAdd(a.taskCounter, 1);
      __syncthreads();
      const int task = __builtin_amdgcn_readfirstlane(taskSh);
      if (task >= a.nTasks) break;
      const ScoreTask tk = a.tasks[task];
      const bool active = fw < tk.nFrames;
      const int nPairs = (tk.nSlots + 1) >> 1;
      STAMP(0);
      // the task's feature rows: one contiguous block, read in order by the whole workgroup
      {
         const float *xs = a.X + (size_t)tk.frame0 * D;
         for (int i = tid; i < tk.nFrames * D; i += 256) xbuf[i] = xs[i];
      }
      // the tile of every state of the task, one per lane (tasks hold at most 64 states); one tile per state: the tile's number is the state's
      int tileV = 0;
      if (lane < tk.nSlots) tileV = a.slotState[tk.slot0 + lane];
      auto pair_tile = [&](int j) { const int k = 2 * j + hsel; return __builtin_amdgcn_readlane(tileV, (k < tk.nSlots ? k : tk.nSlots - 1) & 63); };
      {
         const u4 *W = tab + (size_t)pair_tile(0) * TW4;
#pragma unroll
         for (int j = 0; j < PT; j++)
            if (t7 + 128 * j < TW4) wbuf[0][dst[j]] = W[t7 + 128 * j];
      }
      STAMP(1);
      __syncthreads();
      STAMP(3);

      __builtin_amdgcn_s_setprio(0);                      // (the pairs' loop below runs at a raised priority: gmm_bf16.hip, B16_PRIO)
      // B operand from the rows in LDS: this lane's frame, the 8 k of its k-half in every k-step, scaled, in two fp16 pieces
      h8 zb[KS][2];
      if (active) {
         bool over = false;
         int khL = kh, f = fw + fcol;
         asm volatile("" : "+v"(khL), "+v"(f));         // (the indices below are cheap to recompute per task: hoisted out of the task loop they are spilled)
         if (f > tk.nFrames - 1) f = tk.nFrames - 1;
         const float *row = xbuf + f * D;
#pragma unroll
         for (int ks = 0; ks < KS; ks++) {
            const int c = ks >> 1, i0 = 8 * (ks & 1) + 4 * khL;      // chunk; first of this lane's four dimensions within it
            _Float16 p[2][8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
               int dim = dpc * c + i0 + i;
               const bool pad = i0 + i >= dpc || dim >= D;
               if (pad) dim = D - 1;
               float v = row[dim];
               if (pad) v = 0.0f;
               float v2 = v * v;
               if ((ks & 1) && khL == 1 && i == 3) v2 = 1.0f;      // k = 30 of the chunk: the constant that meets -0.5 sum mu^2 ivar
               v2 *= zsSh[c * 32 + 2 * (i0 + i)]; v *= zsSh[c * 32 + 2 * (i0 + i) + 1];
               over = over || !(v2 <= F16_MAX) || !(fabsf(v) <= F16_MAX);
               split2(v2, p[0][2 * i], p[1][2 * i]);
               split2(v, p[0][2 * i + 1], p[1][2 * i + 1]);
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
               u4 w;
               w[0] = pack2(p[s][0], p[s][1]); w[1] = pack2(p[s][2], p[s][3]);
               w[2] = pack2(p[s][4], p[s][5]); w[3] = pack2(p[s][6], p[s][7]);
               zb[ks][s] = __builtin_bit_cast(h8, w);
            }
            __builtin_amdgcn_sched_barrier(0);               // one k-step's values in flight at a time: the registers are wanted for zb
         }
         if (over) atomicOr(a.rangeFlag, HTKAMD_F16_EFEAT);
      }
      STAMP(0); STAMP_COUNT(8, 1);

      int buf = 0;
      float *o = a.out + tk.outBase + (size_t)(tk.outSlot0 + kh) * tk.ldo + fw + fcol;      // this lane's state (kh of the pair) and frame
      const size_t oStep = 2 * (size_t)tk.ldo;
#if F16W_PIPE
      float yP[16];
#pragma unroll
      for (int r = 0; r < 16; r++) yP[r] = 0.0f;
#endif
      __builtin_amdgcn_s_setprio(1);
      for (int j = 0; j < nPairs; j++) {
         u4 stg[PT];
         const bool more = j + 1 < nPairs;
         if (more) {
            const u4 *W = tab + (size_t)pair_tile(j + 1) * TW4;
#pragma unroll
            for (int q = 0; q < PT; q++)
               if (t7 + 128 * q < TW4) stg[q] = W[t7 + 128 * q];
         }
         STAMP(1); STAMP_COUNT(7, 1);
         if (active) {
            STAMP_COUNT(9, 1);
#if F16W_PIPE
            // The log-sum-exp of the pair BEFORE this one (its 16 values per lane were left in yP) in 18 slices, one behind each of this
            // pair's matrix instructions and fenced there: ~4 vector instructions fit in the shadow of a 32-cycle matrix instruction.
            float m8[8], m4[4], m2[2], mx = 0.0f, e[16], sm = 0.0f, lg = 0.0f, resP = 0.0f;
            auto lse_slice = [&](int sl) {
               if (sl < 2) { for (int r = 4 * sl; r < 4 * sl + 4; r++) m8[r] = fmaxf(yP[r], yP[r + 8]); }
               else if (sl == 2) { for (int r = 0; r < 4; r++) m4[r] = fmaxf(m8[r], m8[r + 4]); }
               else if (sl == 3) { m2[0] = fmaxf(m4[0], m4[1]); m2[1] = fmaxf(m4[2], m4[3]); mx = fmaxf(m2[0], m2[1]); }
               else if (sl < 12) { for (int r = 2 * (sl - 4); r < 2 * (sl - 4) + 2; r++) e[r] = EXP2(yP[r] - mx); }
               else if (sl < 14) { for (int r = 4 * (sl - 12); r < 4 * (sl - 12) + 4; r++) e[r] += e[r + 8]; }
               else if (sl == 14) { for (int r = 0; r < 4; r++) e[r] += e[r + 4]; }
               else if (sl == 15) { sm = (e[0] + e[1]) + (e[2] + e[3]); }
               else if (sl == 16) { lg = LOG2(sm); }
               else { resP = (mx + lg) * 0.69314718055994531f; }
            };
            f16v Cx, Cc;
#pragma unroll
            for (int r = 0; r < 16; r++) { Cx[r] = 0.0f; Cc[r] = 0.0f; }
            h8 wa[KS][2];
            wa[0][0] = __builtin_bit_cast(h8, wbuf[buf][0 * 64 + lane]);
            wa[0][1] = __builtin_bit_cast(h8, wbuf[buf][1 * 64 + lane]);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
               if (ks + 1 < KS) {
                  wa[ks + 1][0] = __builtin_bit_cast(h8, wbuf[buf][((ks + 1) * 2 + 0) * 64 + lane]);
                  wa[ks + 1][1] = __builtin_bit_cast(h8, wbuf[buf][((ks + 1) * 2 + 1) * 64 + lane]);
               }
               __builtin_amdgcn_sched_barrier(0);
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][1], zb[ks][0], Cc, 0, 0, 0);
               if (3 * ks + 0 < 18) lse_slice(3 * ks + 0);
               __builtin_amdgcn_sched_barrier(0);
               Cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][0], zb[ks][0], Cx, 0, 0, 0);
               if (3 * ks + 1 < 18) lse_slice(3 * ks + 1);
               __builtin_amdgcn_sched_barrier(0);
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][0], zb[ks][1], Cc, 0, 0, 0);
               if (3 * ks + 2 < 18) lse_slice(3 * ks + 2);
               __builtin_amdgcn_sched_barrier(0);
            }
            for (int sl = 3 * KS; sl < 18; sl++) lse_slice(sl);      // (fewer than 6 k-steps: the rest of the slices)
            asm volatile("" : "+v"(resP));                       // (computed HERE, not inside the branch of its store)
            if (j > 0 && fw + fcol < tk.nFrames) *o = resP;      // (the pair before always has both its states)
            if (j > 0) o += oStep;
            // this pair's 16 components per lane, registers 4b + r: left for the next round
#pragma unroll
            for (int b = 0; b < 4; b++) {
               const f4 ci = __builtin_bit_cast(f4, wbuf[buf][KS * 128 + kh * 4 + b]);
#pragma unroll
               for (int r = 0; r < 4; r++) yP[4 * b + r] = __builtin_fmaf(Cc[4 * b + r], 1.0f / F16_CORR, Cx[4 * b + r]) + ci[r];
            }
#else
            f16v Cx, Cc;
#pragma unroll
            for (int r = 0; r < 16; r++) { Cx[r] = 0.0f; Cc[r] = 0.0f; }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
               const h8 wa0 = __builtin_bit_cast(h8, wbuf[buf][(ks * 2 + 0) * 64 + lane]), wa1 = __builtin_bit_cast(h8, wbuf[buf][(ks * 2 + 1) * 64 + lane]);
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, zb[ks][0], Cc, 0, 0, 0);
               Cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, zb[ks][0], Cx, 0, 0, 0);
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, zb[ks][1], Cc, 0, 0, 0);
            }
            // this lane's state: its 16 components in registers 4b + r
            float y[16];
#pragma unroll
            for (int b = 0; b < 4; b++) {
               const f4 ci = __builtin_bit_cast(f4, wbuf[buf][KS * 128 + kh * 4 + b]);
#pragma unroll
               for (int r = 0; r < 4; r++) y[4 * b + r] = __builtin_fmaf(Cc[4 * b + r], 1.0f / F16_CORR, Cx[4 * b + r]) + ci[r];
            }
            float m8[8], m4[4];
#pragma unroll
            for (int r = 0; r < 8; r++) m8[r] = fmaxf(y[r], y[r + 8]);
#pragma unroll
            for (int r = 0; r < 4; r++) m4[r] = fmaxf(m8[r], m8[r + 4]);
            const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; r++) e[r] = EXP2(y[r] - mx);
#pragma unroll
            for (int r = 0; r < 8; r++) e[r] += e[r + 8];
#pragma unroll
            for (int r = 0; r < 4; r++) e[r] += e[r + 4];
            const float sm = (e[0] + e[1]) + (e[2] + e[3]);
            const float res = (mx + LOG2(sm)) * 0.69314718055994531f;
            if (fw + fcol < tk.nFrames && 2 * j + kh < tk.nSlots) *o = res;
            o += oStep;
#endif
         }
         STAMP(2);
         if (more) {
#pragma unroll
            for (int q = 0; q < PT; q++)
               if (t7 + 128 * q < TW4) wbuf[buf ^ 1][dst[q]] = stg[q];
         }
         STAMP(4);
         __syncthreads();
         STAMP(5);
         buf ^= 1;
      }
#if F16W_PIPE
      if (active) {                                    // the last pair's log-sum-exp
         float m8[8], m4[4];
#pragma unroll
         for (int r = 0; r < 8; r++) m8[r] = fmaxf(yP[r], yP[r + 8]);
#pragma unroll
         for (int r = 0; r < 4; r++) m4[r] = fmaxf(m8[r], m8[r + 4]);
         const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
         float e[16];
#pragma unroll
         for (int r = 0; r < 16; r++) e[r] = EXP2(yP[r] - mx);
#pragma unroll
         for (int r = 0; r < 8; r++) e[r] += e[r + 8];
#pragma unroll
         for (int r = 0; r < 4; r++) e[r] += e[r + 4];
         const float sm = (e[0] + e[1]) + (e[2] + e[3]);
         if (fw + fcol < tk.nFrames && 2 * (nPairs - 1) + kh < tk.nSlots) *o = (mx + LOG2(sm)) * 0.69314718055994531f;
      }
#endif
   }
   STAMP_FLUSH;
}

int htkamd_launch_score_f16(const htkamd_model *m, const ScoreArgs &a0, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop)
{
   if (a0.nTasks <= 0) return HTKAMD_OK;
   if (!m->d_f16Tab) { htkamd_set_error("score_f16: vector size %d not supported by the fp16 matrix-core path (up to 45)", m->D); return HTKAMD_EMODEL; }
   if (m->f16Stale) {                                // parameters were re-estimated on the device since the table was built (and this path was not in use then)
      int rc = htkamd_model_refresh_f16_device((htkamd_model *)m, (void *)stream);
      if (rc) return rc;
   }
   ((htkamd_model *)m)->fastUse |= HTKAMD_SCORE_F16;
   ScoreArgs a = a0;
   a.f16Tab = m->d_f16Tab; a.f16Ctl = (const int *)m->d_f16Ctl;
   if (!a.rangeFlag) a.rangeFlag = (int *)m->d_f16Ctl + F16_CTL_STICKY;      // the model's sticky flag
   HIPCHECK(hipMemsetAsync(a.taskCounter, 0, sizeof(int) * (a.rangeFlag == a.taskCounter + 1 ? 2 : 1), stream));      // a pass's own flag lies behind its counter
   if (m->f16Wide) {                                 // every state in one tile: 32 x 32 blocks, states in pairs
      int blocks = a.nTasks;
      if (blocks > 256 * F16W_EU) blocks = 256 * F16W_EU;
      dim3 grid(blocks), block(256);
      switch (m->bf16NC) {
      case 3: hipExtLaunchKernelGGL((k_score_f16w<3>), grid, block, 0, stream, evStart, evStop, 0, a); break;
      case 2: hipExtLaunchKernelGGL((k_score_f16w<2>), grid, block, 0, stream, evStart, evStop, 0, a); break;
      case 1: hipExtLaunchKernelGGL((k_score_f16w<1>), grid, block, 0, stream, evStart, evStop, 0, a); break;
      default: htkamd_set_error("score_f16: no kernel for %d K-chunks", m->bf16NC); return HTKAMD_EMODEL;
      }
      HIPCHECK(hipGetLastError());
      return HTKAMD_OK;
   }
   const int parts = B16_TASK_FRAMES / (16 * F16_COL_TILES * F16_WPB);
   int blocks = a.nTasks * parts;
   if (blocks > 256 * ((4 * F16_EU) / F16_WPB)) blocks = 256 * ((4 * F16_EU) / F16_WPB);      // persistent blocks, one task (128 frames x up to 64 states) at a time
   dim3 grid(blocks), block(64 * F16_WPB);
   switch (m->bf16NC) {
   case 3: hipExtLaunchKernelGGL((k_score_f16<3>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 2: hipExtLaunchKernelGGL((k_score_f16<2>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 1: hipExtLaunchKernelGGL((k_score_f16<1>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   default: htkamd_set_error("score_f16: no kernel for %d K-chunks", m->bf16NC); return HTKAMD_EMODEL;
   }
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ the A-operand table, built on the device
struct F16TabArgs {
   int D, NC, S, G;
   const int *stateCompOff, *stateTileOff, *compGauss, *tileState;
   const float *mean, *ivar, *gconst, *compLogWt;
   unsigned short *tab;        // [tile][ 2*NC*64*8 f16 | 64*4 f32 ]
   float *ctl;                 // the model's control block
};

// ranges: per k the largest |coefficient| over the Gaussians and an estimate of the largest |z| (|x| <= |mu| + 8 sigma).  A block reads
// R = 256 / D whole rows at a time, thread t always dimension t % D: its running maxima stay in registers, the block's go through LDS
// to the control block with one atomic each.  The chunks' constants -0.5 sum mu^2 ivar are bounded by the sum of the dimensions' maxima.
#define F16_RANGE_BLOCKS 512
__global__ __launch_bounds__(256) void k_f16_range(F16TabArgs a)
{
   __shared__ unsigned int sm[5 * 48];              // [aq | al | zq | zl | mu^2 ivar][D <= 45]
   const int D = a.D, tid = threadIdx.x, R = 256 / D;
   const float L2E = 1.4426950408889634f;
   for (int i = tid; i < 5 * 48; i += 256) sm[i] = 0u;
   __syncthreads();
   const int d = tid % D;
   float aq = 0.0f, al = 0.0f, zq = 0.0f, zl = 0.0f, mq = 0.0f;
   if (tid < R * D)
      for (size_t g0 = (size_t)blockIdx.x * R; g0 < (size_t)a.G; g0 += (size_t)gridDim.x * R) {
         const size_t idx = g0 * D + tid;
         if (idx >= (size_t)a.G * D) break;
         const float m = a.mean[idx], v = a.ivar[idx];
         const float xm = fabsf(m) + 8.0f * (v > 0.0f ? rsqrtf(v) : 0.0f);
         aq = fmaxf(aq, 0.5f * v * L2E); al = fmaxf(al, fabsf(m * v * L2E)); zq = fmaxf(zq, xm * xm); zl = fmaxf(zl, xm); mq = fmaxf(mq, 0.5f * m * m * v * L2E);
      }
   atomicMax(&sm[d], __float_as_uint(aq)); atomicMax(&sm[48 + d], __float_as_uint(al));
   atomicMax(&sm[96 + d], __float_as_uint(zq)); atomicMax(&sm[144 + d], __float_as_uint(zl)); atomicMax(&sm[192 + d], __float_as_uint(mq));
   __syncthreads();
   unsigned int *rng = (unsigned int *)a.ctl + F16_CTL_RANGE;
   if (tid < D) {
      const int dpc = (D + a.NC - 1) / a.NC, ch = tid / dpc, k = ch * 32 + 2 * (tid - ch * dpc);
      atomicMax(&rng[k], sm[tid]); atomicMax(&rng[k + 1], sm[48 + tid]);
      atomicMax(&rng[96 + k], sm[96 + tid]); atomicMax(&rng[96 + k + 1], sm[144 + tid]);
      atomicMax(&rng[F16_CTL_MQ - F16_CTL_RANGE + tid], sm[192 + tid]);
   }
}

// scale of k: a power of two that puts the largest scaled coefficient and the largest scaled z at the same height
__global__ void k_f16_scale(float *ctl, int D, int NC)
{
   const int k = threadIdx.x;
   if (k >= 96) return;
   const unsigned int *rng = (const unsigned int *)ctl + F16_CTL_RANGE;
   float *scl = ctl;
   float am = __uint_as_float(rng[k]), zm = __uint_as_float(rng[96 + k]);
   if ((k & 31) == 30 && (k >> 5) < NC) {            // the chunk's constant: at most the sum of its dimensions' largest 0.5 mu^2 ivar, against 1
      const int dpc = (D + NC - 1) / NC, dlo = dpc * (k >> 5), dhi = (dlo + dpc < D) ? dlo + dpc : D;
      am = 0.0f;
      for (int d = dlo; d < dhi; d++) am += __uint_as_float(rng[F16_CTL_MQ - F16_CTL_RANGE + d]);
      zm = 1.0f;
   }
   float e = 0.0f;
   if (am > 0.0f && zm > 0.0f) e = rintf(0.5f * (log2f(zm) - log2f(am)));
   scl[k] = exp2f(e);
   scl[96 + k] = exp2f(-e);
}

// The table of a tile: every coefficient in two pieces, and the components' accumulator starts.
//   What two fp16 pieces leave of a coefficient, rho_k = a_k s_k - a1 - a2 2^-11 (|rho| <= 2^-22 |a s|), is the same in every frame: an
// error of the score that does not average out along a path (measured on the bench set: 6e-6 rms per Gaussian over its own frames,
// worst 2e-5 -- occupancies off by several 1e-5).  Its EXPECTATION under the Gaussian itself, sum_k rho_k E[z_k] / s_k with
// E[x^2] = mu^2 + sigma^2, E[x] = mu, E[1] = 1, is a constant of the component and goes into its accumulator start (fp32, added by the
// vector unit): what is left has mean zero over the Gaussian's own frames (3e-7 rms, worst 1e-6 in the same measurement), and the
// chunk constants' share is removed exactly.
template <bool WIDE>
__global__ void k_build_f16tab(F16TabArgs a, int nTiles)
{
   // a workgroup per tile, a thread per (chunk, group of 8 k, component): 64 NC threads, consecutive components in consecutive threads
   // (their 16-byte stores are adjacent); the 4 NC partial sums of a component's correction meet in LDS and are added in a fixed order
   __shared__ double part[16][12];
   const int NC = a.NC, D = a.D;
   const int t = blockIdx.x, r = threadIdx.x, ch = r >> 6, kg = (r >> 4) & 3, rowc = r & 15;
   const int s = a.tileState[t], c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
   const int c = c0 + 16 * (t - a.stateTileOff[s]) + rowc;
   // WIDE: [k-step 2 NC][piece 2][k-half 2][component 16][8 f16] then 16 f32; else [piece 2][chunk NC][lane 64][8 f16] then [lane 64][4 f32]
   const size_t tileShorts = WIDE ? ((size_t)NC * 2 * 2 * 32 + 4) * 8 : (size_t)2 * NC * 64 * 8 + 64 * 8;
   unsigned short *T = a.tab + (size_t)t * tileShorts;
   const bool live = c < c1 && (c1 - c0 == 1 || a.compLogWt[c] > (float)LMINMIX);
   const double L2E = 1.4426950408889634;
   const float *mu = nullptr, *iv = nullptr;
   if (live) { const int g = a.compGauss[c]; mu = a.mean + (size_t)g * D; iv = a.ivar + (size_t)g * D; }
   const int dpc = (D + NC - 1) / NC, dlo = dpc * ch, dhi = (dlo + dpc < D) ? dlo + dpc : D;     // this chunk's dimensions
   _Float16 p[2][8];
   bool over = false;
   double bias = 0.0;
#pragma unroll
   for (int j = 0; j < 8; j++) {
      const int kk = 8 * kg + j, dim = dlo + (kk >> 1);
      float v = 0.0f;
      double ez = 0.0;                                       // E[z_k] under the Gaussian
      if (live && dim < dhi && kk < 30) {
         const double m = mu[dim], w = iv[dim];
         v = (kk & 1) ? (float)(m * w * L2E) : (float)(-0.5 * w * L2E);
         ez = (kk & 1) ? m : ((w > 0.0) ? m * m + 1.0 / w : 0.0);      // (1/variance 0: a dimension outside the Gaussian's stream, coefficient 0)
      }
      if (live && kk == 30) {                               // against B's constant 1: -0.5 sum mu^2 ivar over the chunk
         double q = 0.0;
         for (int i = dlo; i < dhi; i++) q += (double)mu[i] * mu[i] * iv[i];
         v = (float)(-0.5 * q * L2E); ez = 1.0;
      }
      const float sc = a.ctl[ch * 32 + kk];
      v *= sc;
      if (!(fabsf(v) <= F16_MAX)) over = true;
      split2(v, p[0][j], p[1][j]);
      bias += ((double)v - (double)(float)p[0][j] - (double)(float)p[1][j] * (1.0 / F16_CORR)) * ez / (double)sc;
   }
#pragma unroll
   for (int pc = 0; pc < 2; pc++) {
      u4 w;
      w[0] = pack2(p[pc][0], p[pc][1]); w[1] = pack2(p[pc][2], p[pc][3]);
      w[2] = pack2(p[pc][4], p[pc][5]); w[3] = pack2(p[pc][6], p[pc][7]);
      // k = 32 ch + 8 kg + j: WIDE k-step 2 ch + (kg >> 1), k-half kg & 1; else lane 16 kg + component of chunk ch
      const size_t word = WIDE ? (size_t)((2 * ch + (kg >> 1)) * 2 + pc) * 32 + (kg & 1) * 16 + rowc : (size_t)(pc * NC + ch) * 64 + 16 * kg + rowc;
      *(u4 *)(T + word * 8) = w;
   }
   if (over) atomicOr((int *)a.ctl + F16_CTL_FLAG, HTKAMD_F16_EMODEL);
   part[rowc][ch * 4 + kg] = bias;
   __syncthreads();
   if (ch == 0 && kg == 0) {
      float ci = -1.0e30f;
      if (live) {
         double bsum = 0.0;
         for (int i = 0; i < 4 * NC; i++) bsum += part[rowc][i];
         const double k0 = a.gconst[a.compGauss[c]];
         ci = (float)(((c1 - c0 == 1 ? 0.0 : (double)a.compLogWt[c]) - 0.5 * k0) * L2E + bsum);
      }
      if (WIDE) ((float *)(T + (size_t)NC * 2 * 2 * 32 * 8))[rowc] = ci;
      else {
         float *ciBase = (float *)(T + (size_t)2 * NC * 64 * 8);       // [lane][4]: row 4(l>>4)+r lives in lanes with l>>4 == row/4, register row%4
         for (int j = 0; j < 16; j++) ciBase[((rowc >> 2) * 16 + j) * 4 + (rowc & 3)] = ci;
      }
   }
}

int htkamd_model_refresh_f16_device(htkamd_model *m, void *stream)
{
   hipStream_t s = (hipStream_t)stream;
   if (!m->d_f16Tab) return HTKAMD_OK;
   F16TabArgs t;
   t.D = m->D; t.NC = m->bf16NC; t.S = m->S; t.G = m->G; t.stateCompOff = m->d_stateCompOff; t.stateTileOff = m->d_stateTileOff; t.compGauss = m->d_compGauss; t.tileState = m->d_tileState;
   t.mean = m->d_mean; t.ivar = m->d_ivar; t.gconst = m->d_gconst; t.compLogWt = m->d_compLogWt; t.tab = (unsigned short *)m->d_f16Tab; t.ctl = m->d_f16Ctl;
   HIPCHECK(hipMemsetAsync(m->d_f16Ctl + F16_CTL_RANGE, 0, sizeof(int) * (F16_CTL_MQ + 48 - F16_CTL_RANGE), s));      // ranges, the table's flag, k_f16_range's row; the sticky flag stays
   hipLaunchKernelGGL(k_f16_range, dim3(F16_RANGE_BLOCKS), dim3(256), 0, s, t);
   hipLaunchKernelGGL(k_f16_scale, dim3(1), dim3(128), 0, s, m->d_f16Ctl, m->D, m->bf16NC);
   if (m->nTiles > 0) {
      if (m->f16Wide) hipLaunchKernelGGL(k_build_f16tab<true>, dim3(m->nTiles), dim3(64 * m->bf16NC), 0, s, t, m->nTiles);
      else hipLaunchKernelGGL(k_build_f16tab<false>, dim3(m->nTiles), dim3(64 * m->bf16NC), 0, s, t, m->nTiles);
   }
   HIPCHECK(hipGetLastError());
   m->f16Stale = 0;
   return HTKAMD_OK;
}

// the model's sticky range flag (set by launches that have no flag of their own to raise: htkamd_outp_block_mode), read and cleared
int htkamd_model_f16_flag(htkamd_model *m, void *stream, int *flag)
{
   hipStream_t s = (hipStream_t)stream;
   *flag = 0;
   if (!m->d_f16Ctl) return HTKAMD_OK;
   int *p = (int *)m->d_f16Ctl + F16_CTL_STICKY;
   HIPCHECK(hipMemcpyAsync(flag, p, sizeof(int), hipMemcpyDeviceToHost, s));
   HIPCHECK(hipStreamSynchronize(s));
   if (*flag) HIPCHECK(hipMemsetAsync(p, 0, sizeof(int), s));
   return HTKAMD_OK;
}

extern "C" int htkamd_model_f16_check(htkamd_model *m, void *stream)
{
   if (!m) { htkamd_set_error("model_f16_check: NULL"); return HTKAMD_EINVAL; }
   int flag = 0;
   const int rc = htkamd_model_f16_flag(m, stream, &flag);
   if (rc) return rc;
   if (flag) {
      htkamd_set_error("model_f16_check: the fp16 scoring path met %s%s%s outside its range since the last check: repeat those calls with HTKAMD_SCORE_BF16",
                       (flag & HTKAMD_F16_EMODEL) ? "a model coefficient" : "", (flag & HTKAMD_F16_EMODEL) && (flag & HTKAMD_F16_EFEAT) ? " and " : "", (flag & HTKAMD_F16_EFEAT) ? "a feature value" : "");
      return HTKAMD_ERANGE;
   }
   return HTKAMD_OK;
}
