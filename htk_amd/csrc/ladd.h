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

#endif
