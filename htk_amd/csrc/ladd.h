// ladd.h -- device LAdd (HMath.c:1576-1590) on the table built by htkamd_host_build_ladd_table.
//
//   LAdd(x,y): order so x >= y; diff = y-x; if diff < minLogExp return (x < LSMALL ? LZERO : x);
//              else return x + log(1 + exp(diff))
// log(1+exp(diff)) comes from a degree-10 Taylor row of the interval containing diff (4 intervals per
// unit, |r| <= 1/8): 10 fp64 FMAs + one LDS row instead of the ~60 fp64 instructions of exp()+log().
// Absolute error <= 2e-16 (same size as the glibc exp/log pair the reference uses).
#ifndef HTKAMD_LADD_H
#define HTKAMD_LADD_H
#include <hip/hip_runtime.h>
#include "internal.h"

#define LADD_ROW (LADD_DEG + 1)
#define LADD_TAB_DOUBLES (LADD_NK * LADD_ROW)

// Cooperative copy of the table into LDS (call from every thread of the block, then __syncthreads()).
__device__ __forceinline__ void ladd_table_to_lds(double *lds, const double *__restrict__ g)
{
   for (int i = threadIdx.x; i < LADD_TAB_DOUBLES; i += blockDim.x) lds[i] = g[i];
}

__device__ __forceinline__ double ladd_tab(double x, double y, const double minLogExp, const double *tab)
{
   if (x < y) { const double t = x; x = y; y = t; }
   const double d = y - x;
   if (d < minLogExp) return (x < LSMALL) ? LZERO : x;
   const int k = (int)(-d * (double)LADD_INV_H);
   const double r = d + ((double)k + 0.5) * (1.0 / (double)LADD_INV_H);
   const double *row = tab + k * LADD_ROW;
   double f = row[LADD_DEG];
#pragma unroll
   for (int j = LADD_DEG - 1; j >= 0; j--) f = fma(f, r, row[j]);
   return x + f;
}


// Branch-free form: always evaluates the table row (with d clamped into the table) and selects at the end.
// In the latency-bound recursions this lets the scheduler interleave independent LAdd chains.
__device__ __forceinline__ double ladd_tab_bl(const double x, const double y, const double minLogExp, const double *tab)
{
   const double hi = fmax(x, y), lo = fmin(x, y);
   const double d = lo - hi;
   const bool skip = d < minLogExp;
   const double dc = skip ? minLogExp : d;
   const int k = (int)(-dc * (double)LADD_INV_H);
   const double r = dc + ((double)k + 0.5) * (1.0 / (double)LADD_INV_H);
   const double *row = tab + k * LADD_ROW;
   double f = row[LADD_DEG];
#pragma unroll
   for (int j = LADD_DEG - 1; j >= 0; j--) f = fma(f, r, row[j]);
   const double res = hi + f;
   const double keep = (hi < LSMALL) ? LZERO : hi;
   return skip ? keep : res;
}

// exp(x) for x in [EXP_TAB_MIN, ~0] from a table of exp(-k/8) and a degree-8 Taylor polynomial of the remainder
// (|r| <= 1/8: relative error < 1e-13).  Used for occupation / transition counts, which only have to be good
// to the reference's float accumulators; returns 0 below EXP_TAB_MIN (exp(-100) = 3.7e-44).
#define EXP_TAB_MIN (-100.0)
#define EXP_TAB_N 801
__device__ __forceinline__ void exp_table_to_lds(double *lds)
{
   for (int i = threadIdx.x; i < EXP_TAB_N; i += blockDim.x) lds[i] = exp(-(double)i * 0.125);
}
__device__ __forceinline__ double exp_tab(const double x, const double *etab)
{
   const bool live = x > EXP_TAB_MIN;
   const double xc = live ? fmin(x, 0.125) : 0.0;
   int k = (int)(-xc * 8.0);
   if (k < 0) k = 0;
   const double r = xc + (double)k * 0.125;          // in (-1/8, 1/8]
   double p = 1.0 / 40320.0;
   p = fma(p, r, 1.0 / 5040.0);
   p = fma(p, r, 1.0 / 720.0);
   p = fma(p, r, 1.0 / 120.0);
   p = fma(p, r, 1.0 / 24.0);
   p = fma(p, r, 1.0 / 6.0);
   p = fma(p, r, 0.5);
   p = fma(p, r, 1.0);
   p = fma(p, r, 1.0);
   const double v = etab[k] * p;
   return live ? v : 0.0;
}


// LAdd whose result is rounded to FLOAT right away (the mixture sum of ShStrP/cSOutP keeps LogFloat x).
// x and y are doubles holding float values.  Besides the reference's own cut-off (diff < minLogExp) there is an exact
// shortcut: for hi <= -16 and diff < -16 the increment log(1+exp(diff)) < 1.2e-7 is below a quarter of the float
// spacing at hi (>= 2^-19 for |hi| >= 16, and >= 2^-20 on the side towards zero at a binade edge), so
// float(hi + increment) == hi and the table row need not be evaluated.
__device__ __forceinline__ float ladd_tab_f(const float xf, const float yf, const double minLogExp, const double *tab)
{
   const float hi = fmaxf(xf, yf), lo = fminf(xf, yf);
   // (the float difference first: it is the double one to 2^-24 of its size, so below -23.03 the reference's cut-off, minLogExp = -23.0259,
   //  holds for certain and no double is formed)
   // (the pre-test is valid only while the cut-off it anticipates is not below it: minLogExp = -log(-LZERO) = -23.0259 in every model this library makes,
   //  htkamd_host_min_log_exp; a build with another LZERO has to move the literal)
   static_assert(LZERO == -1.0E10, "ladd_tab_f: the float pre-test -23.03 assumes minLogExp = -log(1e10) = -23.0259");
   if (lo - hi < -23.03f) return (hi < (float)LSMALL) ? (float)LZERO : hi;
   const double d = (double)lo - (double)hi;
   if (d < minLogExp) return (hi < (float)LSMALL) ? (float)LZERO : hi;
   if (d < -16.0 && hi <= -16.0f) return hi;
   const int k = (int)(-d * (double)LADD_INV_H);
   const double r = d + ((double)k + 0.5) * (1.0 / (double)LADD_INV_H);
   const double *row = tab + k * LADD_ROW;
   double f = row[LADD_DEG];
#pragma unroll
   for (int j = LADD_DEG - 1; j >= 0; j--) f = fma(f, r, row[j]);
   return (float)((double)hi + f);
}

// Two independent LAdds of a lane (k_score_exact: its two frames) with ONE evaluation of the table row where that is enough.  A lane
// needs the row for a term in one case in ten, but some lane of the 64 nearly always does, so two calls of ladd_tab_f run the eleven
// double FMAs twice per Gaussian with a handful of lanes enabled: here the lanes that need it for either term evaluate it once (for the
// first such term), and the few that need it for both go round again.  Results identical to ladd_tab_f term by term.
__device__ __forceinline__ void ladd_tab_f2(float &a0, const float y0, float &a1, const float y1, const double minLogExp, const double *tab)
{
   const float hi0 = fmaxf(a0, y0), lo0 = fminf(a0, y0), hi1 = fmaxf(a1, y1), lo1 = fminf(a1, y1);
   const float keep0 = (hi0 < (float)LSMALL) ? (float)LZERO : hi0, keep1 = (hi1 < (float)LSMALL) ? (float)LZERO : hi1;
   const double d0 = (double)lo0 - (double)hi0, d1 = (double)lo1 - (double)hi1;
   // row wanted: past the float pre-test, the reference's cut-off and the exact shortcut of ladd_tab_f
   bool n0 = !(lo0 - hi0 < -23.03f) && !(d0 < minLogExp) && !(d0 < -16.0 && hi0 <= -16.0f);
   bool n1 = !(lo1 - hi1 < -23.03f) && !(d1 < minLogExp) && !(d1 < -16.0 && hi1 <= -16.0f);
   // what a term without the row comes to: the cut-off's value, or hi (the shortcut)
   float r0 = (lo0 - hi0 < -23.03f || d0 < minLogExp) ? keep0 : hi0, r1 = (lo1 - hi1 < -23.03f || d1 < minLogExp) ? keep1 : hi1;
   while (n0 || n1) {                                      // (at most two rounds)
      const bool first = n0;
      const double d = first ? d0 : d1;
      const float hi = first ? hi0 : hi1;
      const int k = (int)(-d * (double)LADD_INV_H);
      const double r = d + ((double)k + 0.5) * (1.0 / (double)LADD_INV_H);
      const double *row = tab + k * LADD_ROW;
      double f = row[LADD_DEG];
#pragma unroll
      for (int j = LADD_DEG - 1; j >= 0; j--) f = fma(f, r, row[j]);
      const float v = (float)((double)hi + f);
      if (first) { r0 = v; n0 = false; } else { r1 = v; n1 = false; }
   }
   a0 = r0; a1 = r1;
}

// Tolerance-class forms for the HERest recursions (scoreMode bit HTKAMD_SCORE_FASTLADD): the increment log(1+exp(d)) and the
// occupation exponentials on the hardware's fp32 transcendentals (v_exp_f32 / v_log_f32), the running values stay fp64.
// Absolute error of the increment ~1e-7 (d <= 0: e in (0,1], 1+e rounded to float, v_log_f32 to ~1 ulp), against a bar of 1e-4
// relative on alpha/beta and on the re-estimated parameters.  Branch-free: LZERO operands need no test (exp2(-1.4e10) = 0), and
// the reference's cut-off d < minLogExp = -23.03 is where 1+e rounds to 1 anyway.
__device__ __forceinline__ double ladd_fast(const double x, const double y)
{
   const double hi = fmax(x, y), lo = fmin(x, y);
   const float d = (float)(lo - hi);
   const float e = __builtin_amdgcn_exp2f(d * 1.44269504088896341f);
   const float f = __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994531f;
   return hi + (double)f;
}
__device__ __forceinline__ double exp_fast(const double x)
{
   return (double)__builtin_amdgcn_exp2f((float)x * 1.44269504088896341f);
}
template <bool FAST> __device__ __forceinline__ double ladd_sel(const double x, const double y, const double mle, const double *tab)
{
   if constexpr (FAST) return ladd_fast(x, y); else return ladd_tab(x, y, mle, tab);
}
template <bool FAST> __device__ __forceinline__ double exp_sel(const double x, const double *etab)
{
   if constexpr (FAST) return exp_fast(x); else return exp_tab(x, etab);
}

#endif
