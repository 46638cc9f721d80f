// mfma_rate.hip -- microbenchmark: v_mfma_f32_16x16x4_f32 issue rate on gfx950 with 1/2/4/8 independent accumulator
// chains, at 1..4 waves per SIMD; and the same with VALU filler between MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CH, int FILL>
__global__ void k(float *out, int iters)
{
   f4 c[8];
   for (int i = 0; i < 8; i++) c[i] = (f4){0, 0, 0, 0};
   float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
   float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
#pragma unroll
         for (int j = 0; j < CH; j++) {
            c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
            if (FILL) {
#pragma unroll
               for (int f = 0; f < FILL; f++)
                  asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(v0) : "v"(v1), "v"(v2));
            }
         }
      }
   }
   float s = v0 + v3;
   for (int i = 0; i < CH; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH, int FILL> static void run()
{
   float *out;
   hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   const int iters = 500;
   for (int wps : {1, 2, 4}) {
      int blocks = 256 * wps;
      hipLaunchKernelGGL((k<CH, FILL>), dim3(blocks), dim3(256), 0, 0, out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<CH, FILL>), dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double nm = (double)blocks * 4 * iters * 16 * CH;       // MFMAs
      double tf = nm * 2048 / (ms * 1e-3) / 1e12;
      double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 16 * CH);   // cycles per MFMA per SIMD at 2.4 GHz
      printf("chains=%d fill=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  %.1f cyc/MFMA/SIMD\n", CH, FILL, wps, ms, tf, cyc);
   }
   hipFree(out);
}

int main()
{
   run<1, 0>(); run<2, 0>(); run<4, 0>(); run<8, 0>();
   run<4, 2>(); run<4, 4>(); run<4, 6>(); run<4, 8>();
   return 0;
}
