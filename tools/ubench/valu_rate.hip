// valu_rate.hip -- microbenchmark: issue rate of scalar vs packed FP32 VALU (non-FMA) on gfx950,
// and of FP64 add/fma, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int MODE>
__global__ void k(float *out, int iters, float s)
{
   float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
   typedef float v2 __attribute__((ext_vector_type(2)));
   v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
   double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < REP; r++) {
         if (MODE == 0) {      // 8 independent v_mul_f32 with an SGPR operand
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
         } else if (MODE == 1) {   // 4 independent v_pk_mul_f32 (8 floats)
            asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
         } else if (MODE == 2) {   // 4 independent v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
         } else if (MODE == 3) {   // 4 v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
         } else if (MODE == 4) {   // 4 v_add_f64
            asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
         } else if (MODE == 5) {   // 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %8, %0, %1\n v_fma_f32 %1, %8, %1, %2\n v_fma_f32 %2, %8, %2, %3\n v_fma_f32 %3, %8, %3, %4\n"
                         "v_fma_f32 %4, %8, %4, %5\n v_fma_f32 %5, %8, %5, %6\n v_fma_f32 %6, %8, %6, %7\n v_fma_f32 %7, %8, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
         } else if (MODE == 6) {   // 4 v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
         }
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
}

template <int MODE> static void run(const char *name, int instrPerRep, int lanesOps)
{
   float *out;
   hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   const int iters = 2000;
   for (int wavesPerSimd : {1, 2, 4, 8}) {
      int threads = 256;                          // 4 waves per block = 1 per SIMD
      int blocks = 256 * wavesPerSimd;            // 256 CUs
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0001f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double instr = (double)iters * REP * instrPerRep;           // wave-instructions per wave
      double waveInstrPerSimdPerSec = instr * wavesPerSimd / (ms * 1e-3);
      double cyc = 2.4e9 / waveInstrPerSimdPerSec;                // cycles per wave-instruction per SIMD at 2.4 GHz nominal
      double tops = instr * 64.0 * lanesOps * blocks * 4 / (ms * 1e-3) / 1e12;
      printf("%-14s waves/SIMD=%d  %.3f ms  %.2f cyc/instr/SIMD(@2.4GHz)  %.1f Tlane-op/s\n", name, wavesPerSimd, ms, cyc, tops);
   }
   hipFree(out);
}

int main()
{
   run<0>("v_mul_f32", 8, 1);
   run<1>("v_pk_mul_f32", 4, 2);
   run<2>("v_pk_add_f32", 4, 2);
   run<5>("v_fma_f32", 8, 1);
   run<6>("v_pk_fma_f32", 4, 2);
   run<3>("v_fma_f64", 4, 1);
   run<4>("v_add_f64", 4, 1);
   return 0;
}
