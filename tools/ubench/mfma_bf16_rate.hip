// mfma_bf16_rate.hip -- microbenchmark: v_mfma_f32_16x16x32_bf16 issue rate on gfx950 with 1/2/4/8 independent accumulator chains at
// 1..4 waves per SIMD, and with FILL independent VALU instructions after every MFMA (does vector issue hide behind the matrix pipe?).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int CH, int FILL, int TRANS>
__global__ void k(float *out, int iters)
{
   f4 c[8];
   for (int i = 0; i < 8; i++) c[i] = (f4){0, 0, 0, 0};
   bf8 a, b;
   for (int i = 0; i < 8; i++) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + threadIdx.x * 1e-4f); }
   float v0 = threadIdx.x * 1e-3f, v1 = 1.0f + threadIdx.x * 1e-4f, v2 = v0 + v1, v3 = v0 - v1;
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
#pragma unroll
         for (int j = 0; j < CH; j++) {
            c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL; f++) {
               if (TRANS) asm volatile("v_exp_f32 %0, %1\n" : "=v"(v3) : "v"(v1));
               else asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(v0) : "v"(v1), "v"(v2));
            }
         }
      }
   }
   float s = v0 + v3;
   for (int i = 0; i < CH; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH, int FILL, int TRANS> static void run()
{
   float *out;
   hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   const int iters = 500;
   for (int wps = 1; wps <= 4; wps++) {          // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD; wps blocks per CU
      hipLaunchKernelGGL((k<CH, FILL, TRANS>), dim3(256 * wps), dim3(256), 0, 0, out, 10);
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<CH, FILL, TRANS>), dim3(256 * wps), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mf = (double)iters * 16 * CH * wps;                // MFMAs per SIMD
      printf("chains %d fill %d%s waves/SIMD %d: %.1f cycles per MFMA per SIMD (2.4 GHz)\n", CH, FILL, TRANS ? "exp" : "fma", wps, ms * 1e-3 * 2.4e9 / mf);
   }
   hipFree(out);
}

int main()
{
   run<1, 0, 0>(); run<2, 0, 0>(); run<4, 0, 0>(); run<8, 0, 0>();
   run<2, 1, 0>(); run<2, 2, 0>(); run<2, 4, 0>(); run<4, 2, 0>(); run<4, 4, 0>();
   run<2, 1, 1>(); run<4, 1, 1>();
   return 0;
}
