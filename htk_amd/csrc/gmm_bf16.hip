// gmm_bf16.hip -- K1b: GMM state log-likelihoods on the BF16 matrix pipe with three-way operand splitting.
//
// Same quantity and the same expanded form as gmm_mfma.hip (K1m):
//     log2( w_m N(x; mu_m, var_m) ) = cinit_m + sum_k a_mk z_k ,   z = (x_0^2, x_0, x_1^2, x_1, ...),
//     a_m,2i = -0.5 ivar_mi log2(e),  a_m,2i+1 = mu_mi ivar_mi log2(e)
// K1m contracts it with v_mfma_f32_16x16x4_f32, which runs at the FP32 VECTOR rate (64 FLOP/clk/SIMD) and shares the FP32 lanes with
// every other vector instruction of the SIMD.  The BF16 pipe is 16 times faster per clock (v_mfma_f32_16x16x32_bf16: 16 cycles for a
// 16x16x32 block) and leaves half of its cycles to vector issue, but a bf16 carries 8 significant bits.  So both operands are split,
// exactly, into three bf16 pieces each,  a = a1 + a2 + a3,  z = z1 + z2 + z3  (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2):
// |a - a1| <= 2^-9 |a|, |a - a1 - a2| <= 2^-18 |a|), and the product is summed over the six piece pairs that matter at fp32 accuracy:
//     a z  =  a1 z1 + (a1 z2 + a2 z1) + (a2 z2 + a1 z3 + a3 z1)  +  O(2^-27 |a z|),
// accumulated in fp32 by the matrix unit.  Six bf16 blocks of K = 32 against eight fp32 blocks of K = 4 for the same 32 terms:
// 6 x 16 = 96 cycles instead of 8 x 32 = 256, and the mixture's log-sum-exp (v_exp_f32 / v_log_f32 / row swaps) issues beside them.
// Tolerance class like K1m (|score - reference| <= 1e-3, typically 1e-5): the Viterbi path and the C ABI default stay on gmm_exact.hip.
//
// Accuracy (round 3).  The matrix unit rounds its fp32 accumulator after every instruction, at the accumulator's magnitude.  With the
// expanded form's constant -0.5 sum mu^2 ivar (~ -230 in base-2 units at D = 39) as the accumulator's start, all 18 instructions of a
// tile rounded at ulp(256) = 3e-5: |score error| up to 2e-4, re-estimated variances up to 1.8e-4 off the reference's at the headline
// size -- over north_star's 1e-4.  So (a) every K chunk carries COMPLETE squares: the dimensions are dealt out to the chunks (dpc =
// ceil(D/NC) <= 15 per chunk) and k = 30 of a chunk holds, against a constant 1 in B, the chunk's share -0.5 sum_{d in chunk} mu_d^2
// ivar_d, so that a chunk's product is -0.5 sum ivar (x - mu)^2 over its dimensions and the accumulator moves monotonically from
// zero to -0.5 sum ivar (x - mu)^2; (b) the accumulator STARTS AT ZERO: the fifteen correction products (everything but a1 z1) come
// first and stay below ~2, where a rounding is 1e-7, then the three leading products, and log w - 0.5 gConst (~ -55) is added once at
// the end by the vector unit.  Same matrix instruction count; measured at the headline size: see DESIGN.md §2.
//
// Layout.  K is cut into NC chunks of 32 (NC = ceil(D/15): 3 at D = 39); chunk c covers the dimensions dpc*c .. dpc*c + dpc - 1 as
// (x^2, x) pairs at k = 2i, 2i+1, then zeros, the constant at k = 30.  Lane l of a wave holds, per
// chunk c, the 8 consecutive k = 32c + 8(l>>4) + j of its matrix row / column (cdna_hip_programming.md "A/B operand lane maps"):
//   A (Gaussians = rows, l&15): device-built table, per tile of 16 components: [piece 3][chunk NC][lane 64][8 bf16], then the
//       accumulator start (log w - 0.5 gConst) log2(e) as [lane 64][4 f32] in the C layout (row = 4(l>>4) + r, col = l&15: rows only matter).
//   B (frames = columns, l&15): built once per task from the feature rows: 4 consecutive dimensions per lane and chunk, squared and
//       plain, split into the three pieces (3 x NC x 4 VGPRs per 16-frame column tile).
// Task structure, LDS staging of the table (one copy per workgroup, a tile ahead), log-sum-exp and stores are K1m's.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) int cint;

#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define LOG2(x) __builtin_amdgcn_logf(x)
#ifndef B16_COL_TILES
#define B16_COL_TILES 2                                 // 16-frame column tiles per wavefront: 2 (four wavefronts per 128-frame task) or 4 (two)
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

// round-to-nearest-even bf16 of a finite float, as bits, and back
__device__ __host__ __forceinline__ unsigned short bf16_bits(float x)
{
   unsigned int u;
   memcpy(&u, &x, 4);
   u += 0x7FFFu + ((u >> 16) & 1u);
   return (unsigned short)(u >> 16);
}
__device__ __host__ __forceinline__ float bf16_val(unsigned short b)
{
   const unsigned int u = (unsigned int)b << 16;
   float x;
   memcpy(&x, &u, 4);
   return x;
}
// x = p1 + p2 + p3 + O(2^-27 |x|), every piece a bf16 and every difference exact in fp32
__device__ __host__ __forceinline__ void split3(float x, unsigned short &p1, unsigned short &p2, unsigned short &p3)
{
   p1 = bf16_bits(x);
   const float r1 = x - bf16_val(p1);
   p2 = bf16_bits(r1);
   const float r2 = r1 - bf16_val(p2);
   p3 = bf16_bits(r2);
}

#ifndef B16_WPB
#define B16_WPB (8 / B16_COL_TILES)                     // wavefronts per workgroup: together a whole 128-frame task (B16_WPB smaller: the task in parts)
#endif
template <int NC>
__global__ __launch_bounds__(64 * B16_WPB, B16_COL_TILES > 2 ? 2 : 3) void k_score_bf16(ScoreArgs a)      // second figure: wavefronts per SIMD the register budget is cut for
{
   constexpr int NT = 64 * B16_WPB, FPW = 16 * B16_COL_TILES, HALVES = B16_TASK_FRAMES / (FPW * B16_WPB);
   constexpr int TWB = 3 * NC * 64 * 16 + 64 * 16;     // bytes per fragment tile
   constexpr int TW4 = TWB / 16;                       // 16-byte words per tile
   constexpr int PT = (TW4 + NT - 1) / NT;             // words staged per thread
   __shared__ u4 wbuf[2][TW4];
   __shared__ int taskSh;
   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const int col = lane & 15, kg = lane >> 4;
   const int D = a.D;
   const int dpc = (D + NC - 1) / NC;                  // dimensions per K chunk (<= 15)
   const u4 *tab = (const u4 *)a.bf16Tab;

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

      // B operand: this lane's frame (col) of each column tile, the 8 k of lane group kg in every chunk, in three bf16 pieces
      bf8 zb[B16_COL_TILES][3][NC];
      if (active) {
      int kgL = kg, colL = col;
      asm volatile("" : "+v"(kgL), "+v"(colL));         // (the indices below are cheap to recompute per task: hoisted out of the task loop they are spilled)
#pragma unroll
      for (int ft = 0; ft < B16_COL_TILES; ft++) {
         int f = fw + ft * 16 + colL;
         if (f > tk.nFrames - 1) f = tk.nFrames - 1;
         const float *row = a.X + (size_t)(tk.frame0 + f) * D;
#pragma unroll
         for (int c = 0; c < NC; c++) {
            const int d0 = dpc * c + 4 * kgL;         // dimensions d0..d0+3 -> k = 32c + 8kg + (0..7) = (x^2, x) pairs
            unsigned short p[3][8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
               int dim = d0 + i;
               const bool pad = 4 * kgL + i >= dpc || dim >= D;
               if (pad) dim = D - 1;
               float v = row[dim];
               if (pad) v = 0.0f;
               float v2 = v * v;
               if (i == 3 && kgL == 3) v2 = 1.0f;      // k = 30: the constant that meets the chunk's -0.5 sum mu^2 ivar
               split3(v2, p[0][2 * i], p[1][2 * i], p[2][2 * i]);
               split3(v, p[0][2 * i + 1], p[1][2 * i + 1], p[2][2 * i + 1]);
            }
#pragma unroll
            for (int s = 0; s < 3; s++) {
               u4 w;
               w[0] = p[s][0] | ((unsigned int)p[s][1] << 16); w[1] = p[s][2] | ((unsigned int)p[s][3] << 16);
               w[2] = p[s][4] | ((unsigned int)p[s][5] << 16); w[3] = p[s][6] | ((unsigned int)p[s][7] << 16);
               zb[ft][s][c] = __builtin_bit_cast(bf8, w);
            }
         }
      }
      }
      __syncthreads();

      int buf = 0;
      for (int k = 0; k < tk.nSlots; k++) {
         const int t1 = __builtin_amdgcn_readlane(tEndV, k);
         const int nextFirst = (k + 1 < tk.nSlots) ? __builtin_amdgcn_readlane(tFirstV, (k + 1) & 63) : -1;
         float rM[B16_COL_TILES], rS[B16_COL_TILES];
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
               bf8 wa[3][NC];
#pragma unroll
               for (int s = 0; s < 3; s++)
#pragma unroll
                  for (int c = 0; c < NC; c++) wa[s][c] = __builtin_bit_cast(bf8, wbuf[buf][(s * NC + c) * 64 + lane]);
               const f4 ci = __builtin_bit_cast(f4, wbuf[buf][3 * NC * 64 + lane]);
               f4 Cx[B16_COL_TILES];
#pragma unroll
               for (int ft = 0; ft < B16_COL_TILES; ft++) Cx[ft] = (f4)(0.0f);
               // from zero, smallest first: (a2 z2, a1 z3, a3 z1), then (a1 z2, a2 z1) -- all below ~2 --, then the leading products
               // a1 z1, complete squares per chunk; the accumulator start is added last, by the vector unit
#pragma unroll
               for (int c = 0; c < NC; c++)
#pragma unroll
                  for (int ft = 0; ft < B16_COL_TILES; ft++) {
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][c], zb[ft][1][c], Cx[ft], 0, 0, 0);
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][c], zb[ft][2][c], Cx[ft], 0, 0, 0);
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[2][c], zb[ft][0][c], Cx[ft], 0, 0, 0);
                  }
#pragma unroll
               for (int c = 0; c < NC; c++)
#pragma unroll
                  for (int ft = 0; ft < B16_COL_TILES; ft++) {
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][c], zb[ft][1][c], Cx[ft], 0, 0, 0);
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][c], zb[ft][0][c], Cx[ft], 0, 0, 0);
                  }
#pragma unroll
               for (int c = 0; c < NC; c++)
#pragma unroll
                  for (int ft = 0; ft < B16_COL_TILES; ft++)
                     Cx[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][c], zb[ft][0][c], Cx[ft], 0, 0, 0);
               // log-sum-exp over the tile's 16 rows: 4 in this lane, the rest in lanes ^16, ^32, ^48 (base-2 logs, as K1m)
#pragma unroll
               for (int ft = 0; ft < B16_COL_TILES; ft++) {
                  const f4 y = Cx[ft] + ci;
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
         for (int ft = 0; ft < B16_COL_TILES; ft++) {
            const float r = (rM[ft] + LOG2(rS[ft])) * 0.69314718055994531f;
            if (kg == ft) res = r;
         }
         float *o = a.out + tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + fw;
         if (active && lane < FPW && fw + lane < tk.nFrames) o[lane] = res;
      }
   }
}

// ------------------------------------------------------------------------------------ the 32 x 32 form (round 4)
// K1bw `k_score_bf16w`: K1h's shape (gmm_f16.hip, k_score_f16w) under K1b's arithmetic.  Sets whose states all have <= 16 components
// (one tile per state: the headline set).  v_mfma_f32_32x32x16_bf16 on PAIRS of states -- the components of state A on the accumulator
// rows whose bit 2 is clear, those of state B on the others -- so that a lane ends with all 16 components of ONE state for ITS frame:
// the mixture's log-sum-exp is a tree inside the lane (no row swaps), cut into slices that issue behind the next pair's matrix
// instructions; the task's 128 feature rows come through LDS as one contiguous block.  Three bf16 pieces per operand, six products:
//     Cx += a1 z1                                      (16-bit products: the matrix unit adds them without dropping a bit)
//     Cc += a2 z2 + a1 z3 + a3 z1 + a1 z2 + a2 z1      (everything below 2^-8 of the leading term: an accumulator that stays small)
//     y   = (Cx + Cc) + (log w - 0.5 gConst) log2(e)   (by the vector unit, last)
// with K1b's complete squares per K chunk.  36 matrix instructions of 32 cycles per pair against K1h's 18: twice the matrix time under
// the same vector, LDS and staging work -- which therefore hides better (K1b, the 16 x 16 form: 1.22 ms; this: see DESIGN.md §4).
//   Table per tile: [k-step 2 NC][piece 3][k-half 2][component 16][8 bf16], then (log w - 0.5 gConst) log2(e) [16 f32].  In LDS a pair
//   is [k-step][piece][lane 64][8 bf16] with lane = 32 k-half + 8 (comp >> 2) + 4 h + (comp & 3), then the two states' constants.
typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef B16W_EU
#define B16W_EU 2
#endif
// KS k-steps of 16.  Even KS: NC = KS / 2 chunks of 32 as described above.  KS = 5, the DENSE layout for 31 <= D <= 39 (39: the usual
// MFCC_E_D_A, whose 2 D + 2 = 80 terms fill five k-steps exactly where three chunks take six -- a sixth of the matrix instructions,
// table words and LDS traffic less): two blocks of dimensions, D0 = ceil(D / 2) and D - D0, each as (x^2, x) pairs followed by its
// constant -0.5 sum mu^2 ivar (k = 2 D0 and k = 2 D + 1); k-steps cut across the pairs, which the products do not mind.  The accumulator
// still moves from zero through complete squares; between the constants up to 20 dimensions are open instead of 8 (measured: DESIGN §4).
// Round 5: the constants sit side by side in ONE pair slot behind the first block, k = 2 D0 and 2 D0 + 1 with D0 = floor(D / 2): every other
// slot is the (x^2, x) pair of a dimension at an even k, which lets the kernel build its operand pair by pair -- one packed conversion per
// piece (v_cvt_pk_bf16_f32) and a dimension number that is the pair's number less one behind the constants.  The per-slot arithmetic of the
// first layout (selects on the slot's kind and block) was 1 700 vector instructions per task and wavefront against 890 in the pairs' loop,
// on the issue port the matrix instructions share (profiles/README.md r05b).  Open dimensions between the constants: 20 at most, as before
// (the second block's constant comes before its dimensions instead of after them).
// Round 6: WHERE the accumulator of the leading products walks.  With both constants in the middle (pair floor(D / 2), round 5) it climbs
// to +1/2 sum_{d < 19} mu^2 ivar (~ +110 in base-2 units on the headline set), drops to ~ -100 and climbs back: four of its five roundings
// happen at ulp(64) = 7.6e-6, and the matrix unit cuts every product at the accumulator's last bit.  B16_LAYOUT 2: the accumulator STARTS at
// the share of the first 16 dimensions (the pairs of the first two k-steps), I = -1/2 sum_{d < 16} mu^2 ivar, read from the tile beside
// log w - 1/2 gConst; the constants' pair sits at pair PC = floor(4 D / 5) (the fourth k-step) and holds the shares of dimensions 16 .. PC - 1
// and PC .. D - 1.  The walk is then  I -> ~I/2 -> -Q16 (complete squares) -> ~+I/2 -> ~-I/2 -> -Q : never beyond half of what it was.
// B16_LAYOUT 0 keeps round 5's placement (no start value, constants at pair floor(D / 2)) for comparisons.
#ifndef B16_LAYOUT
#define B16_LAYOUT 2
#endif
__device__ __host__ __forceinline__ int dense_D0(int D) { return B16_LAYOUT == 2 ? (4 * D) / 5 : D >> 1; }      // the constants' pair
__device__ __host__ __forceinline__ int dense_NI(int D) { return B16_LAYOUT == 2 ? 16 : 0; }                   // dimensions whose share the accumulator starts from
__device__ __forceinline__ void dense_slot(int k, int D, int &dim, int &kind)      // kind 0: x^2, 1: x, 2: first constant, 3: second constant, -1: padding
{
   const int D0 = dense_D0(D);
   const int j = k >> 1;                               // the pair
   dim = j - (j > D0 ? 1 : 0);
   kind = k & 1;
   if (j == D0) kind = 2 + (k & 1);
   if (dim >= D && j != D0) kind = -1;
   if (kind >= 2 || kind < 0) dim = 0;
}
// the dimensions a constant stands for: which = 0 the accumulator's start, 1 / 2 the first / second constant of the pair
__device__ __forceinline__ void dense_const_range(int which, int D, int &lo, int &hi)
{
   const int NI = dense_NI(D), D0 = dense_D0(D);
   lo = which == 0 ? 0 : which == 1 ? NI : D0;
   hi = which == 0 ? NI : which == 1 ? D0 : D;
   if (hi < lo) hi = lo;
}

// workgroups per CU by registers and LDS: six k-steps 180 registers / 60 KB -> 2; five 168 / 51 KB -> 3 (measured against 2: DESIGN §4); four 152 / 40 KB -> 3; two 111 / 20 KB -> 4
#ifndef B16W_EU5
#define B16W_EU5 3
#endif
constexpr int b16w_eu(int KS) { return KS >= 6 ? B16W_EU : KS == 5 ? B16W_EU5 : KS == 4 ? 3 : 4; }
#ifndef B16_PRIO_LEVEL
#define B16_PRIO_LEVEL 1
#endif
#ifndef B16_PRIO
#define B16_PRIO 1                                      /* 1: the pairs' loop at a raised wavefront priority, the operand build at the normal one: a workgroup in its products goes
                                                           before one that is building (-3 %: 0.94 -> 0.91 ms; levels 1 - 3 alike).  Experiments: 2 the log-sum-exp behind the products at a
                                                           lowered priority (0 %), 3 the priority dropped around every slice of it (as 1) */
#endif
#ifndef B16_ABL
#define B16_ABL 0                                       /* diagnostic builds: 1 no barrier per pair, 2 no log-sum-exp, 4 one fragment load per pair, 8 no staging */
#endif
#ifdef B16_CLK                                          // cycle stamps of thread 0 of every workgroup, summed: task fetch | rows + first pair landed | operand built | pairs' loop
__device__ unsigned long long g_b16clk[8];
#define B16_STAMP(i_) do { if (tid == 0) { const unsigned long long c_ = __builtin_readcyclecounter(); clkAcc[i_] += c_ - clk0; clk0 = c_; } } while (0)
#else
#define B16_STAMP(i_) do { } while (0)
#endif
template <int KS>
__global__ __launch_bounds__(256, b16w_eu(KS)) void k_score_bf16w(ScoreArgs a)
{
#ifdef B16_CLK
   unsigned long long clkAcc[5] = {0, 0, 0, 0, 0}, clk0 = __builtin_readcyclecounter();
#endif
   static_assert(B16_TASK_FRAMES == 128, "four wavefronts x 32 frames");
   constexpr bool DENSE = (KS & 1) != 0;
   constexpr int NC = (KS + 1) / 2;
   constexpr int CW = DENSE ? 8 : 4;                   // 16-byte words of per-component constants in a tile: log w - 0.5 gConst; dense: then the accumulator's start
   constexpr int TW4 = KS * 3 * 32 + CW;               // 16-byte words per tile in the table
   constexpr int PW4 = KS * 3 * 64 + 2 * CW;           // ... per pair in LDS
   __shared__ u4 wbuf[2][PW4];
   __shared__ float xbuf[128 * (DENSE ? 8 * KS : 15 * NC)];      // the task's 128 feature rows (D <= 15 NC; dense: 2 D + 2 <= 16 KS), as they lie in memory
   __shared__ int taskSh;
   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const int fcol = lane & 31, kh = lane >> 5;
   const int D = a.D;
   const int dpc = (D + NC - 1) / NC;                  // dimensions per K chunk (<= 15)
   const u4 *tab = (const u4 *)a.bf16Tab;
   // staging by LDS-DMA (global_load_lds_dwordx4: no registers, no LDS store instructions): a row (k-step, piece) of a PAIR is 64 x 16 bytes
   // in LDS, one wave-instruction; the destination is wave-uniform base + 16 lane, so the interleave of the two states' tiles is made on
   // the SOURCE side -- LDS lane l' = 32 k-half + 8 (comp >> 2) + 4 h + (comp & 3) takes word (row 32 + 16 k-half + comp) of tile h.
   // Wavefront w brings rows w, w + 4, ...; the last one also the two states' constants (8 words).
   const int hL = (lane >> 2) & 1;
   const int srcOff = (lane >> 5) * 16 + ((lane >> 3) & 3) * 4 + (lane & 3);
#define GLDS16(src_, dst_) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_), (__attribute__((address_space(3))) void *)(dst_), 16, 0, 0)
   const int fw = 32 * wv;                             // this wave's first frame in the task's tile

   // eight task queues, one per XCD (ScoreArgs::qStart): a workgroup pulls from the queue of the XCD it runs on (HW_REG_XCC_ID) and moves on
   // round the others when that one is empty -- the frame tiles of a state chunk then meet their 64 table tiles in ONE L2
   int myQ = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);
   for (;;) {
      if (tid == 0) {
         int t_;
         if (a.qStart) {
            t_ = 0x7fffffff;
            for (int k = 0; k < 8; k++) {
               const int tk_ = atomicAdd(a.qCounters + myQ, 1);
               if (tk_ < a.qStart[myQ + 1] - a.qStart[myQ]) { t_ = a.qStart[myQ] + tk_; break; }
               myQ = (myQ + 1) & 7;
            }
         } else t_ = atomicAdd(a.taskCounter, 1);
         taskSh = t_;
      }
      __syncthreads();
      const int task = __builtin_amdgcn_readfirstlane(taskSh);
      if (task >= a.nTasks) break;
      const ScoreTask tk = a.tasks[task];
      B16_STAMP(0);
      const bool active = fw < tk.nFrames;
      const int nPairs = (tk.nSlots + 1) >> 1;
      {
         const float *xs = a.X + (size_t)tk.frame0 * D;
         for (int i = tid; i < tk.nFrames * D; i += 256) xbuf[i] = xs[i];
      }
      int tileV = 0;
      if (lane < tk.nSlots) tileV = a.slotState[tk.slot0 + lane];
      auto pair_tile = [&](int j, int h) { const int k = 2 * j + h; return __builtin_amdgcn_readlane(tileV, (k < tk.nSlots ? k : tk.nSlots - 1) & 63); };
      auto stage_pair = [&](int j, int bufi) {
         const int tA = pair_tile(j, 0), tB = pair_tile(j, 1);
         // (a uniform base and ONE 32-bit offset per lane: the table is < 4 GB -- 64-bit lane addresses were spilled)
         const unsigned int off = ((unsigned int)(hL ? tB : tA) * TW4 + srcOff) * 16u;
#pragma unroll
         for (int r0 = 0; r0 < KS * 3; r0 += 4) {
            const int r = __builtin_amdgcn_readfirstlane(r0 + wv);
            if (r < KS * 3) GLDS16((const char *)tab + (off + (unsigned int)r * 512u), &wbuf[bufi][r * 64]);
         }
         // the two states' constants: LDS words [A's four | B's four], dense: then [A's start values | B's]
         if (wv == 3 && lane < 2 * CW)
            GLDS16((const char *)tab + (((unsigned int)(((lane >> 2) & 1) ? tB : tA) * TW4 + KS * 96 + (lane >> 3) * 4 + (lane & 3)) * 16u), &wbuf[bufi][KS * 192]);
      };
      stage_pair(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      B16_STAMP(1);

      if (B16_PRIO) __builtin_amdgcn_s_setprio(0);
      // B operand from the rows in LDS: this lane's frame, the 8 k of its k-half in every k-step, in three bf16 pieces
      bf8 zb[KS][3];
      if (active) {
         int khL = kh, f = fw + fcol;
         asm volatile("" : "+v"(khL), "+v"(f));         // (cheap to recompute per task: hoisted out of the task loop they are spilled)
         if (f > tk.nFrames - 1) f = tk.nFrames - 1;
         const float *row = xbuf + f * D;
#pragma unroll
         for (int ks = 0; ks < KS; ks++) {
            unsigned short p[3][8];
            if constexpr (DENSE) {
               // this lane's four pairs of the k-step: pair j = 8 ks + 4 k-half + q holds dimension j (before the constants' pair D0), the two
               // constants, or dimension j - 1; a pair's two values are split together (dense_slot is the same map, slot by slot)
               const int D0 = dense_D0(D), jb = 8 * ks + 4 * khL;
               typedef float v2f_ __attribute__((ext_vector_type(2)));
               typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
               unsigned int w3[3][4];
#pragma unroll
               for (int q = 0; q < 4; q++) {
                  const int j = jb + q;
                  const int dim = j - (j > D0 ? 1 : 0);
                  float x = row[dim];                              // (past the row's end for a padding pair: inside xbuf, and not used)
                  x = (j == D0) ? 1.0f : (dim >= D ? 0.0f : x);
                  v2f_ v = {x * x, x};
#pragma unroll
                  for (int s = 0; s < 3; s++) {
                     const unsigned int u = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf2_));
                     w3[s][q] = u;
                     if (s < 2) v = v - (v2f_){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
                  }
               }
#pragma unroll
               for (int s = 0; s < 3; s++) {
                  u4 w; w[0] = w3[s][0]; w[1] = w3[s][1]; w[2] = w3[s][2]; w[3] = w3[s][3];
                  zb[ks][s] = __builtin_bit_cast(bf8, w);
               }
               __builtin_amdgcn_sched_barrier(0);
               continue;
            } else {
            const int c = ks >> 1, i0 = 8 * (ks & 1) + 4 * khL;      // chunk; first of this lane's four dimensions within it
#pragma unroll
            for (int i = 0; i < 4; i++) {
               int dim = dpc * c + i0 + i;
               const bool pad = i0 + i >= dpc || dim >= D;
               if (pad) dim = D - 1;
               float v = row[dim];
               if (pad) v = 0.0f;
               float v2 = v * v;
               if ((ks & 1) && khL == 1 && i == 3) v2 = 1.0f;      // k = 30 of the chunk: the constant that meets -0.5 sum mu^2 ivar
               split3(v2, p[0][2 * i], p[1][2 * i], p[2][2 * i]);
               split3(v, p[0][2 * i + 1], p[1][2 * i + 1], p[2][2 * i + 1]);
            }
            }
#pragma unroll
            for (int s = 0; s < 3; s++) {
               u4 w;
               w[0] = p[s][0] | ((unsigned int)p[s][1] << 16); w[1] = p[s][2] | ((unsigned int)p[s][3] << 16);
               w[2] = p[s][4] | ((unsigned int)p[s][5] << 16); w[3] = p[s][6] | ((unsigned int)p[s][7] << 16);
               zb[ks][s] = __builtin_bit_cast(bf8, w);
            }
            __builtin_amdgcn_sched_barrier(0);
         }
      }

      B16_STAMP(2);
      int buf = 0;
      float *oPrev = a.out + tk.outBase + (size_t)(tk.outSlot0 + kh) * tk.ldo + fw + fcol;      // this lane's state (kh of the pair) and frame: the pair BEFORE the round's
      const size_t oStep = 2 * (size_t)tk.ldo;
      oPrev -= oStep;
      float *oQ = oPrev;
      float yP[16];
#pragma unroll
      for (int r = 0; r < 16; r++) yP[r] = 0.0f;
      float resQ = 0.0f;                                // the result of the pair before the last one: stored at the top of the next round, so
      bool haveQ = false;                               // that the store is long done where the round's staging is waited for (vmcnt counts both)
      bool prevAct = false;                             // the pair before this round's was computed: its sums are in yP
      // the last pair's log-sum-exp and that of a pair behind which this wavefront sits a round out: the same maxima, the same tree of sums as the slices
      auto lse_full = [&]() {
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
         return (mx + LOG2(sm)) * 0.69314718055994531f;
      };
      // Setotprob's ranges per chain state (HFB.c:1014 with 1177 / 1215; ScoreArgs::slotRange): a wavefront whose 32 frames lie outside the
      // ranges of BOTH states of a pair leaves the pair's products, log-sum-exp and store out -- it stages, and waits at the barrier
      int rLoV = 0x7fffffff, rHiV = -1;
      if (a.slotRange && lane < tk.nSlots) { const int2 rr = ((const int2 *)a.slotRange)[tk.slot0 + lane]; rLoV = rr.x; rHiV = rr.y; }
      const int wRow0 = tk.frame0 + fw, wRow1 = wRow0 + 31;
      for (int j = 0; j < nPairs; j++) {
         const bool more = j + 1 < nPairs;
         bool actJ = active;
         if (a.slotRange) {
            const int lo0 = __builtin_amdgcn_readlane(rLoV, (2 * j) & 63), lo1 = __builtin_amdgcn_readlane(rLoV, (2 * j + 1) & 63);
            const int hi0 = __builtin_amdgcn_readlane(rHiV, (2 * j) & 63), hi1 = __builtin_amdgcn_readlane(rHiV, (2 * j + 1) & 63);
            actJ = active && (lo0 < lo1 ? lo0 : lo1) <= wRow1 && (hi0 > hi1 ? hi0 : hi1) >= wRow0;
         }
         if (haveQ) { if (fw + fcol < tk.nFrames) *oQ = resQ; haveQ = false; }
         if (more && !(B16_ABL & 8)) stage_pair(j + 1, buf ^ 1);
         if (!actJ) {
            if (prevAct) { resQ = lse_full(); oQ = oPrev; haveQ = true; }
         } else {
            // the log-sum-exp of the pair BEFORE this one (left in yP) in 18 slices, written between the matrix instructions of this pair (the
            // compiler places them: pinning every slice with sched_barrier cost 16 registers and 1 % at three workgroups per CU)
            // (the sum's tree is ((e_k + e_k+8) + (e_k+4 + e_k+12)) for k = 0 .. 3, then (E0 + E1) + (E2 + E3): the exponentials are taken in
            //  that order and added as they come -- four partial sums alive instead of sixteen terms)
            // (round 5: the matrix instructions share the vector ALU's issue port -- 4.9 vector instructions per matrix instruction, 16 of them
            //  transcendental, left the matrix pipe 58 % busy; diagnostic builds (B16_ABL): without this log-sum-exp the kernel takes 0.77 ms,
            //  its floor with nothing but the matrix instructions 0.76, with it 0.93.  Maxima three at a time, differences and sums as packed pairs)
            typedef float v2f __attribute__((ext_vector_type(2)));
            float m6[6], m2[2], mx = 0.0f, sm = 0.0f, lg = 0.0f, resP = 0.0f;
            v2f nmx = {0.0f, 0.0f}, eA = {0.0f, 0.0f}, sA = {0.0f, 0.0f}, sB = {0.0f, 0.0f}, E01 = {0.0f, 0.0f}, E23 = {0.0f, 0.0f};
            // the exponentials of components (2 i, 2 i + 1) -- neighbours in the registers, so that the packed instructions need no moves
            auto exp_pair = [&](int i) { const v2f d = (v2f){yP[2 * i], yP[2 * i + 1]} + nmx; return (v2f){EXP2(d.x), EXP2(d.y)}; };
            auto lse_slice = [&](int sl) {
#if (B16_ABL & 2)
               if (sl == 17) resP = yP[0] + yP[5];
               return;
#endif
               if (sl == 0) { for (int r = 0; r < 3; r++) m6[r] = fmaxf(fmaxf(yP[3 * r], yP[3 * r + 1]), yP[3 * r + 2]); }
               else if (sl == 1) { for (int r = 3; r < 5; r++) m6[r] = fmaxf(fmaxf(yP[3 * r], yP[3 * r + 1]), yP[3 * r + 2]); m6[5] = yP[15]; }
               else if (sl == 2) { m2[0] = fmaxf(fmaxf(m6[0], m6[1]), m6[2]); m2[1] = fmaxf(fmaxf(m6[3], m6[4]), m6[5]); }
               else if (sl == 3) { mx = fmaxf(m2[0], m2[1]); nmx = (v2f){-mx, -mx}; }
               // the sum's tree, as at the task's last pair below: s_k = e_k + e_k+8, E_k = s_k + s_k+4 (k = 0 .. 3), (E0 + E1) + (E2 + E3) --
               // two components at a time: (s0,s1) = (e0,e1) + (e8,e9); (s4,s5) likewise; (E0,E1) = (s0,s1) + (s4,s5); then components 2, 3
               else if (sl == 4) eA = exp_pair(0);
               else if (sl == 5) sA = eA + exp_pair(4);
               else if (sl == 6) eA = exp_pair(2);
               else if (sl == 7) { sB = eA + exp_pair(6); E01 = sA + sB; }
               else if (sl == 8) eA = exp_pair(1);
               else if (sl == 9) sA = eA + exp_pair(5);
               else if (sl == 10) eA = exp_pair(3);
               else if (sl == 11) { sB = eA + exp_pair(7); E23 = sA + sB; }
               else if (sl < 15) { }
               else if (sl == 15) { sm = (E01.x + E01.y) + (E23.x + E23.y); }
               else if (sl == 16) { lg = LOG2(sm); }
               else { resP = (mx + lg) * 0.69314718055994531f; }
            };
            f16v Cx, Cc;
#pragma unroll
            for (int r = 0; r < 16; r++) { Cx[r] = 0.0f; Cc[r] = 0.0f; }      // (no instructions: the first product of each takes the constant 0)
            auto cx_start = [&]() {
               if constexpr (DENSE && B16_LAYOUT == 2) {                     // the leading products start from the share of the first 16 dimensions (dense_NI)
#pragma unroll
                  for (int b = 0; b < 4; b++) {
                     const f4 c0 = __builtin_bit_cast(f4, wbuf[buf][KS * 192 + 8 + kh * 4 + b]);
#pragma unroll
                     for (int r = 0; r < 4; r++) Cx[4 * b + r] = c0[r];
                  }
               }
            };
#ifndef B16_INIT_LATE
#define B16_INIT_LATE 1                                 /* the start values are read behind the round's first products (their product is the sixth) */
#endif
            if (!B16_INIT_LATE) cx_start();
            bf8 wa[KS][3];
#pragma unroll
            for (int s = 0; s < 3; s++) wa[0][s] = __builtin_bit_cast(bf8, wbuf[buf][(0 * 3 + s) * 64 + lane]);
            if (B16_PRIO) __builtin_amdgcn_s_setprio(B16_PRIO == 2 ? 2 : B16_PRIO_LEVEL);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
               if (ks + 1 < KS) {
#pragma unroll
                  for (int s = 0; s < 3; s++) wa[ks + 1][s] = (B16_ABL & 4) ? wa[0][s] : __builtin_bit_cast(bf8, wbuf[buf][((ks + 1) * 3 + s) * 64 + lane]);
               }
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][1], zb[ks][1], Cc, 0, 0, 0);
               if (B16_PRIO != 2 && 3 * ks + 0 < 18) { if (B16_PRIO == 3) __builtin_amdgcn_s_setprio(0); lse_slice(3 * ks + 0); if (B16_PRIO == 3) __builtin_amdgcn_s_setprio(B16_PRIO_LEVEL); }
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][0], zb[ks][2], Cc, 0, 0, 0);
               if (B16_INIT_LATE && ks == 0) cx_start();
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][2], zb[ks][0], Cc, 0, 0, 0);
               if (B16_PRIO != 2 && 3 * ks + 1 < 18) { if (B16_PRIO == 3) __builtin_amdgcn_s_setprio(0); lse_slice(3 * ks + 1); if (B16_PRIO == 3) __builtin_amdgcn_s_setprio(B16_PRIO_LEVEL); }
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][0], zb[ks][1], Cc, 0, 0, 0);
               Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][1], zb[ks][0], Cc, 0, 0, 0);
               if (B16_PRIO != 2 && 3 * ks + 2 < 18) { if (B16_PRIO == 3) __builtin_amdgcn_s_setprio(0); lse_slice(3 * ks + 2); if (B16_PRIO == 3) __builtin_amdgcn_s_setprio(B16_PRIO_LEVEL); }
               Cx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][0], zb[ks][0], Cx, 0, 0, 0);
            }
            if (B16_PRIO == 2) { __builtin_amdgcn_s_setprio(0); for (int sl = 0; sl < 18; sl++) lse_slice(sl); }
            else
            for (int sl = 3 * KS; sl < 18; sl++) lse_slice(sl);      // (fewer than 6 k-steps: the rest of the slices)
            asm volatile("" : "+v"(resP));
            resQ = resP; oQ = oPrev; haveQ = prevAct;    // (the pair before always has both its states)
#pragma unroll
            for (int b = 0; b < 4; b++) {
               const f4 ci = __builtin_bit_cast(f4, wbuf[buf][KS * 192 + kh * 4 + b]);
#pragma unroll
               for (int r = 0; r < 4; r++) yP[4 * b + r] = (Cx[4 * b + r] + Cc[4 * b + r]) + ci[r];
            }
         }
         prevAct = actJ; oPrev += oStep;
#if !(B16_ABL & 1)
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next pair's rows have landed (issued a whole round ago)
         __syncthreads();
#endif
         buf ^= 1;
      }
      B16_STAMP(3);
#ifdef B16_CLK
      if (tid == 0) clkAcc[4] += (unsigned long long)nPairs;
#endif
      if (haveQ) { if (fw + fcol < tk.nFrames) *oQ = resQ; }
      if (prevAct) {                                   // the last pair's log-sum-exp
         const float res = lse_full();
         if (fw + fcol < tk.nFrames && 2 * (nPairs - 1) + kh < tk.nSlots) *oPrev = res;
      }
   }
#ifdef B16_CLK
   if (tid == 0) { for (int i = 0; i < 5; i++) atomicAdd(&g_b16clk[i], clkAcc[i]); atomicAdd(&g_b16clk[5], 1ull); }
#endif
}

int htkamd_launch_score_bf16(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop)
{
   if (a.nTasks <= 0) return HTKAMD_OK;
   if (!m->d_bf16Tab) { htkamd_set_error("score_bf16: vector size %d not supported by the bf16 matrix-core path (up to 45)", m->D); return HTKAMD_EMODEL; }
   if (m->bf16Stale) {                               // parameters were re-estimated on the device since the table was built (and this path was not in use then)
      int rc = htkamd_model_refresh_bf16_device((htkamd_model *)m, (void *)stream);
      if (rc) return rc;
   }
   ((htkamd_model *)m)->fastUse |= HTKAMD_SCORE_BF16;
   if (a.qCounters == a.taskCounter + 8) HIPCHECK(hipMemsetAsync(a.taskCounter, 0, 16 * sizeof(int), stream));      // the counter and the eight queues' behind it: one fill
   else {
      HIPCHECK(hipMemsetAsync(a.taskCounter, 0, sizeof(int), stream));
      if (a.qCounters) HIPCHECK(hipMemsetAsync(a.qCounters, 0, 8 * sizeof(int), stream));
   }
   if (m->f16Wide && (size_t)m->nTiles * ((size_t)3 * m->bf16NC * 64 * 16 + 64 * 16) >= ((size_t)1 << 32)) {
      htkamd_set_error("score_bf16: a table of %d tiles is beyond the kernel's 32-bit staging offsets", m->nTiles); return HTKAMD_EMODEL;
   }
   if (m->f16Wide) {                                 // every state in one tile: 32 x 32 blocks, states in pairs (the table is in that layout)
      dim3 block(256);
#define W_LAUNCH(KS_) do { const int b_ = a.nTasks < 256 * b16w_eu(KS_) ? a.nTasks : 256 * b16w_eu(KS_); \
                           hipExtLaunchKernelGGL((k_score_bf16w<KS_>), dim3(b_), block, 0, stream, evStart, evStop, 0, a); } while (0)
      if (m->bf16Dense) {
         W_LAUNCH(5); HIPCHECK(hipGetLastError());
#ifdef B16_CLK
         {  unsigned long long h[8]; static int nth = 0;
            HIPCHECK(hipStreamSynchronize(stream)); HIPCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_b16clk), sizeof(h)));
            if (++nth % 8 == 0) fprintf(stderr, "k_score_bf16w<5> cycles summed over %llu workgroups (cumulative): fetch %llu | landed %llu | operand %llu | pairs %llu ; pairs %llu, tasks %d\n", h[5], h[0], h[1], h[2], h[3], h[4], a.nTasks); }
#endif
         return HTKAMD_OK;
      }
      switch (m->bf16NC) {
      case 3: W_LAUNCH(6); break;
      case 2: W_LAUNCH(4); break;
      case 1: W_LAUNCH(2); break;
      default: htkamd_set_error("score_bf16: no kernel for %d K-chunks", m->bf16NC); return HTKAMD_EMODEL;
      }
#undef W_LAUNCH
      HIPCHECK(hipGetLastError());
      return HTKAMD_OK;
   }
   const int parts = B16_TASK_FRAMES / (16 * B16_COL_TILES * B16_WPB);
   int blocks = a.nTasks * parts;
   if (blocks > 256 * ((B16_COL_TILES > 2 ? 8 : 12) / B16_WPB)) blocks = 256 * ((B16_COL_TILES > 2 ? 8 : 12) / B16_WPB);      // persistent blocks, one task (128 frames x up to 64 states) or half-task at a time
   dim3 grid(blocks), block(64 * B16_WPB);
   switch (m->bf16NC) {
   case 3: hipExtLaunchKernelGGL((k_score_bf16<3>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 2: hipExtLaunchKernelGGL((k_score_bf16<2>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 1: hipExtLaunchKernelGGL((k_score_bf16<1>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   default: htkamd_set_error("score_bf16: no kernel for %d K-chunks", m->bf16NC); return HTKAMD_EMODEL;
   }
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ the A-operand table, built on the device
struct Bf16TabArgs {
   int D, NC, S;
   const int *stateCompOff, *stateTileOff, *compGauss, *tileState;
   const float *mean, *ivar, *gconst, *compLogWt;
   unsigned short *tab;        // [tile][ 3*NC*64*8 bf16 | 64*4 f32 ]
};

// one thread per (tile, chunk, lane): its 8 coefficients in three pieces = three 16-byte stores, consecutive lanes to consecutive
// words; the 16 threads (chunk 0, lane group 0) of a tile also write their component's accumulator start
template <bool WIDE>
__global__ void k_build_bf16tab(Bf16TabArgs a, int nTiles)
{
   const int NC = a.NC, D = a.D;
   const int idx = blockIdx.x * blockDim.x + threadIdx.x;
   if (idx >= nTiles * NC * 64) return;
   const int t = idx / (NC * 64), r = idx - t * (NC * 64), ch = r >> 6, lane = r & 63, rowc = lane & 15, kg = lane >> 4;
   const int s = a.tileState[t], c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
   const int c = c0 + 16 * (t - a.stateTileOff[s]) + rowc;
   // WIDE: [k-step 2 NC][piece 3][k-half 2][component 16][8 bf16] then 16 f32; else [piece 3][chunk NC][lane 64][8 bf16] then [lane 64][4 f32]
   const size_t tileShorts = WIDE ? ((size_t)NC * 2 * 3 * 32 + 4) * 8 : (size_t)3 * NC * 64 * 8 + 64 * 8;      // the f32 part counted in shorts
   unsigned short *T = a.tab + (size_t)t * tileShorts;
   const bool live = c < c1 && (c1 - c0 == 1 || a.compLogWt[c] > (float)LMINMIX);
   const double L2E = 1.4426950408889634;
   const float *mu = nullptr, *iv = nullptr;
   if (live) { const int g = a.compGauss[c]; mu = a.mean + (size_t)g * D; iv = a.ivar + (size_t)g * D; }
   const int dpc = (D + NC - 1) / NC, dlo = dpc * ch, dhi = (dlo + dpc < D) ? dlo + dpc : D;     // this chunk's dimensions
   unsigned short p[3][8];
#pragma unroll
   for (int j = 0; j < 8; j++) {
      const int kk = 8 * kg + j, dim = dlo + (kk >> 1);
      float v = 0.0f;
      if (live && dim < dhi) v = (kk & 1) ? (float)((double)mu[dim] * iv[dim] * L2E) : (float)(-0.5 * (double)iv[dim] * L2E);
      if (live && kk == 30) {                               // against B's constant 1: -0.5 sum mu^2 ivar over the chunk
         double q = 0.0;
         for (int i = dlo; i < dhi; i++) q += (double)mu[i] * mu[i] * iv[i];
         v = (float)(-0.5 * q * L2E);
      }
      split3(v, p[0][j], p[1][j], p[2][j]);
   }
#pragma unroll
   for (int pc = 0; pc < 3; pc++) {
      u4 w;
      w[0] = p[pc][0] | ((unsigned int)p[pc][1] << 16); w[1] = p[pc][2] | ((unsigned int)p[pc][3] << 16);
      w[2] = p[pc][4] | ((unsigned int)p[pc][5] << 16); w[3] = p[pc][6] | ((unsigned int)p[pc][7] << 16);
      // k = 32 ch + 8 kg + j: WIDE k-step 2 ch + (kg >> 1), k-half kg & 1
      const size_t word = WIDE ? (size_t)((2 * ch + (kg >> 1)) * 3 + pc) * 32 + (kg & 1) * 16 + rowc : (size_t)(pc * NC + ch) * 64 + lane;
      *(u4 *)(T + word * 8) = w;
   }
   if (ch == 0 && kg == 0) {
      float ci = -1.0e30f;
      if (live) {
         const double k0 = a.gconst[a.compGauss[c]];
         ci = (float)(((c1 - c0 == 1 ? 0.0 : (double)a.compLogWt[c]) - 0.5 * k0) * L2E);
      }
      if (WIDE) ((float *)(T + (size_t)NC * 2 * 3 * 32 * 8))[rowc] = ci;
      else {
         float *ciBase = (float *)(T + (size_t)3 * NC * 64 * 8);       // [lane][4]: row 4(l>>4)+r lives in lanes with l>>4 == row/4, register row%4
         for (int j = 0; j < 16; j++) ciBase[((rowc >> 2) * 16 + j) * 4 + (rowc & 3)] = ci;
      }
   }
}

// the DENSE layout's table (k_score_bf16w<5>): one thread per (tile, k-step, k-half, component): its 8 coefficients in three pieces
__global__ void k_build_bf16tab_dense(Bf16TabArgs a, int nTiles, int KS)
{
   const int D = a.D;
   const int idx = blockIdx.x * blockDim.x + threadIdx.x;
   if (idx >= nTiles * KS * 32) return;
   const int t = idx / (KS * 32), r = idx - t * (KS * 32), ks = r >> 5, khf = (r >> 4) & 1, rowc = r & 15;
   const int s = a.tileState[t], c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
   const int c = c0 + 16 * (t - a.stateTileOff[s]) + rowc;
   const size_t tileShorts = ((size_t)KS * 3 * 32 + 8) * 8;
   unsigned short *T = a.tab + (size_t)t * tileShorts;
   const bool live = c < c1 && (c1 - c0 == 1 || a.compLogWt[c] > (float)LMINMIX);
   const double L2E = 1.4426950408889634;
   const float *mu = nullptr, *iv = nullptr;
   if (live) { const int g = a.compGauss[c]; mu = a.mean + (size_t)g * D; iv = a.ivar + (size_t)g * D; }
   auto share = [&](int which) {                           // -0.5 sum mu^2 ivar over the dimensions constant `which` stands for (dense_const_range)
      int lo, hi; dense_const_range(which, D, lo, hi);
      double q = 0.0;
      for (int i = lo; i < hi; i++) q += (double)mu[i] * mu[i] * iv[i];
      return (float)(-0.5 * q * L2E);
   };
   unsigned short p[3][8];
#pragma unroll
   for (int j = 0; j < 8; j++) {
      int dim, kind;
      dense_slot(16 * ks + 8 * khf + j, D, dim, kind);
      float v = 0.0f;
      if (live && kind == 0) v = (float)(-0.5 * (double)iv[dim] * L2E);
      else if (live && kind == 1) v = (float)((double)mu[dim] * iv[dim] * L2E);
      else if (live && kind >= 2) v = share(kind - 1);      // against B's constant 1
      split3(v, p[0][j], p[1][j], p[2][j]);
   }
#pragma unroll
   for (int pc = 0; pc < 3; pc++) {
      u4 w;
      w[0] = p[pc][0] | ((unsigned int)p[pc][1] << 16); w[1] = p[pc][2] | ((unsigned int)p[pc][3] << 16);
      w[2] = p[pc][4] | ((unsigned int)p[pc][5] << 16); w[3] = p[pc][6] | ((unsigned int)p[pc][7] << 16);
      *(u4 *)(T + ((size_t)(ks * 3 + pc) * 32 + khf * 16 + rowc) * 8) = w;
   }
   if (ks == 0 && khf == 0) {
      float ci = -1.0e30f;
      if (live) {
         const double k0 = a.gconst[a.compGauss[c]];
         ci = (float)(((c1 - c0 == 1 ? 0.0 : (double)a.compLogWt[c]) - 0.5 * k0) * L2E);
      }
      ((float *)(T + (size_t)KS * 3 * 32 * 8))[rowc] = ci;
      ((float *)(T + (size_t)KS * 3 * 32 * 8))[16 + rowc] = live ? share(0) : 0.0f;      // the accumulator's start
   }
}

// The same table through LDS (round 5): a workgroup takes TB tiles; their 16 TB rows of means and inverse variances arrive in storage order
// (a row is D consecutive floats; k_build_bf16tab_dense read them 8 scattered floats per lane, 32 lines per load), the two constants of a
// row are summed once (by one thread each, in the order of the sum above) instead of by each of the threads whose slots hold them, and the
// slots are cut from LDS.  Same arithmetic per value: the tables are identical.
#define BT_TILES 4
__global__ __launch_bounds__(256) void k_build_bf16tab_dense_lds(Bf16TabArgs a, int nTiles, int KS)
{
   extern __shared__ float btRows[];                         // [BT_TILES*16][2][D] (mu, ivar), then [BT_TILES*16][3] constants (float: start, first, second), then [BT_TILES*16] live flags
   const int D = a.D;
   const int t0 = blockIdx.x * BT_TILES;
   const int nT = (nTiles - t0 < BT_TILES) ? nTiles - t0 : BT_TILES;
   const int nR = nT * 16;
   float *qc = btRows + (size_t)BT_TILES * 16 * 2 * D;
   int *gOf = (int *)(qc + BT_TILES * 16 * 3);               // the row's Gaussian, -1: not live
   int *cOf = gOf + BT_TILES * 16;
   const double L2E = 1.4426950408889634;
   if ((int)threadIdx.x < nR) {
      const int t = t0 + (threadIdx.x >> 4), rowc = threadIdx.x & 15;
      const int s = a.tileState[t], c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
      const int c = c0 + 16 * (t - a.stateTileOff[s]) + rowc;
      const bool live = c < c1 && (c1 - c0 == 1 || a.compLogWt[c] > (float)LMINMIX);
      gOf[threadIdx.x] = live ? a.compGauss[c] : -1;
      cOf[threadIdx.x] = (c1 - c0 == 1) ? -1 - c : c;         // single-component states carry no weight
   }
   __syncthreads();
   for (int i = threadIdx.x; i < nR * D; i += blockDim.x) {
      const int r = i / D, d = i - r * D, g = gOf[r];
      float mu = 0.0f, iv = 0.0f;
      if (g >= 0) { mu = a.mean[(size_t)g * D + d]; iv = a.ivar[(size_t)g * D + d]; }
      btRows[(size_t)(2 * r) * D + d] = mu; btRows[(size_t)(2 * r + 1) * D + d] = iv;
   }
   __syncthreads();
   if ((int)threadIdx.x < 3 * nR) {
      const int r = threadIdx.x / 3, h = threadIdx.x - 3 * r;
      const float *mu = btRows + (size_t)(2 * r) * D, *iv = mu + D;
      int lo, hi; dense_const_range(h, D, lo, hi);
      double q = 0.0;
      for (int i = lo; i < hi; i++) q += (double)mu[i] * mu[i] * iv[i];
      qc[threadIdx.x] = (float)(-0.5 * q * L2E);
   }
   __syncthreads();
   const size_t tileShorts = ((size_t)KS * 3 * 32 + 8) * 8;
   for (int w = threadIdx.x; w < nT * KS * 32; w += blockDim.x) {
      const int tl = w / (KS * 32), r = w - tl * (KS * 32), ks = r >> 5, khf = (r >> 4) & 1, rowc = r & 15;
      const int row = tl * 16 + rowc;
      const bool live = gOf[row] >= 0;
      const float *mu = btRows + (size_t)(2 * row) * D, *iv = mu + D;
      unsigned short *T = a.tab + (size_t)(t0 + tl) * tileShorts;
      unsigned short p[3][8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
         int dim, kind;
         dense_slot(16 * ks + 8 * khf + j, D, dim, kind);
         float v = 0.0f;
         if (live && kind == 0) v = (float)(-0.5 * (double)iv[dim] * L2E);
         else if (live && kind == 1) v = (float)((double)mu[dim] * iv[dim] * L2E);
         else if (live && kind >= 2) v = qc[3 * row + (kind - 1)];
         split3(v, p[0][j], p[1][j], p[2][j]);
      }
#pragma unroll
      for (int pc = 0; pc < 3; pc++) {
         u4 wv;
         wv[0] = p[pc][0] | ((unsigned int)p[pc][1] << 16); wv[1] = p[pc][2] | ((unsigned int)p[pc][3] << 16);
         wv[2] = p[pc][4] | ((unsigned int)p[pc][5] << 16); wv[3] = p[pc][6] | ((unsigned int)p[pc][7] << 16);
         *(u4 *)(T + ((size_t)(ks * 3 + pc) * 32 + khf * 16 + rowc) * 8) = wv;
      }
      if (ks == 0 && khf == 0) {
         float ci = -1.0e30f;
         if (live) {
            const double k0 = a.gconst[gOf[row]];
            const int cc = cOf[row];
            ci = (float)(((cc < 0 ? 0.0 : (double)a.compLogWt[cc]) - 0.5 * k0) * L2E);
         }
         ((float *)(T + (size_t)KS * 3 * 32 * 8))[rowc] = ci;
         ((float *)(T + (size_t)KS * 3 * 32 * 8))[16 + rowc] = live ? qc[3 * row] : 0.0f;      // the accumulator's start
      }
   }
}

int htkamd_model_refresh_bf16_device(htkamd_model *m, void *stream)
{
   hipStream_t s = (hipStream_t)stream;
   if (!m->d_bf16Tab) return HTKAMD_OK;
   Bf16TabArgs t;
   t.D = m->D; t.NC = m->bf16NC; t.S = m->S; t.stateCompOff = m->d_stateCompOff; t.stateTileOff = m->d_stateTileOff; t.compGauss = m->d_compGauss; t.tileState = m->d_tileState;
   t.mean = m->d_mean; t.ivar = m->d_ivar; t.gconst = m->d_gconst; t.compLogWt = m->d_compLogWt; t.tab = (unsigned short *)m->d_bf16Tab;
   const int n = m->nTiles * m->bf16NC * 64;
   const size_t ldsDense = sizeof(float) * ((size_t)BT_TILES * 16 * 2 * m->D + BT_TILES * 16 * 3) + sizeof(int) * BT_TILES * 16 * 2;
   if (m->f16Wide && m->bf16Dense && ldsDense <= 60 * 1024 && !getenv("HTKAMD_TAB_GATHER"))
      hipLaunchKernelGGL(k_build_bf16tab_dense_lds, dim3((m->nTiles + BT_TILES - 1) / BT_TILES), dim3(256), ldsDense, s, t, m->nTiles, 5);
   else if (m->f16Wide && m->bf16Dense) hipLaunchKernelGGL(k_build_bf16tab_dense, dim3((m->nTiles * 5 * 32 + 255) / 256), dim3(256), 0, s, t, m->nTiles, 5);
   else if (m->f16Wide) hipLaunchKernelGGL(k_build_bf16tab<true>, dim3((n + 255) / 256), dim3(256), 0, s, t, m->nTiles);
   else hipLaunchKernelGGL(k_build_bf16tab<false>, dim3((n + 255) / 256), dim3(256), 0, s, t, m->nTiles);
   HIPCHECK(hipGetLastError());
   m->bf16Stale = 0;
   return HTKAMD_OK;
}
