// tools/ubench/mfma_err.hip: rounding behaviour of one matrix instruction (f16 32x32x16, f16 16x16x32, bf16 16x16x32) against float64, with a large
// fp32 accumulator input as the scoring kernels have it.   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_err.hip -o tools/ubench/_bin/mfma_err
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D32, float *D16, float *Db)
{
   const int l = threadIdx.x;
   {  // 32x32x16: A[i][k] row i = l&31, k = 8(l>>5)+j; B[k][n] col n = l&31
      h8 a, b; for (int j = 0; j < 8; j++) { a[j] = A[(l & 31) * 32 + 8 * (l >> 5) + j]; b[j] = B[(l & 31) * 32 + 8 * (l >> 5) + j]; }
      f16v c; for (int r = 0; r < 16; r++) c[r] = C[(8 * (r >> 2) + 4 * (l >> 5) + (r & 3)) * 32 + (l & 31)];
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
      for (int r = 0; r < 16; r++) D32[(8 * (r >> 2) + 4 * (l >> 5) + (r & 3)) * 32 + (l & 31)] = c[r];
   }
   {  // 16x16x32: row/col l&15, k = 8(l>>4)+j (rows/cols 0..15 of the same matrices, K = 32)
      h8 a, b; for (int j = 0; j < 8; j++) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(l & 15) * 32 + 8 * (l >> 4) + j]; }
      f4 c; for (int r = 0; r < 4; r++) c[r] = C[(4 * (l >> 4) + r) * 32 + (l & 15)];
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
      for (int r = 0; r < 4; r++) D16[(4 * (l >> 4) + r) * 32 + (l & 15)] = c[r];
      b8 ab, bb; for (int j = 0; j < 8; j++) { ab[j] = (__bf16)(float)a[j]; bb[j] = (__bf16)(float)b[j]; }
      f4 c2; for (int r = 0; r < 4; r++) c2[r] = C[(4 * (l >> 4) + r) * 32 + (l & 15)];
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c2, 0, 0, 0);
      for (int r = 0; r < 4; r++) Db[(4 * (l >> 4) + r) * 32 + (l & 15)] = c2[r];
   }
}
int main()
{
   const int N = 32 * 32;
   std::vector<_Float16> A(N), B(N); std::vector<float> C(N), D32(N), D16(N), Db(N);
   _Float16 *dA, *dB; float *dC, *d32, *d16, *db;
   hipMalloc(&dA, N * 2); hipMalloc(&dB, N * 2); hipMalloc(&dC, N * 4); hipMalloc(&d32, N * 4); hipMalloc(&d16, N * 4); hipMalloc(&db, N * 4);
   for (int mode = 0; mode < 3; mode++) {
      const double cmag = mode == 0 ? 0.0 : mode == 1 ? 300.0 : 300.0, pmag = mode == 2 ? 0.01 : 4.0;   // accumulator magnitude, operand magnitude
      double e32 = 0, e16 = 0, eb = 0, ulp = 0; int n32 = 0, n16 = 0;
      srand(7 + mode);
      for (int it = 0; it < 200; it++) {
         for (int i = 0; i < N; i++) { A[i] = (_Float16)(pmag * (rand() / (double)RAND_MAX - 0.5)); B[i] = (_Float16)(pmag * (rand() / (double)RAND_MAX - 0.5)); C[i] = (float)(cmag * (rand() / (double)RAND_MAX - 0.5)); }
         hipMemcpy(dA, A.data(), N * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), N * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), N * 4, hipMemcpyHostToDevice);
         hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, d32, d16, db);
         hipMemcpy(D32.data(), d32, N * 4, hipMemcpyDeviceToHost); hipMemcpy(D16.data(), d16, N * 4, hipMemcpyDeviceToHost); hipMemcpy(Db.data(), db, N * 4, hipMemcpyDeviceToHost);
         for (int i = 0; i < 32; i++) for (int n = 0; n < 32; n++) {
            double x = C[i * 32 + n]; for (int kk = 0; kk < 16; kk++) x += (double)(float)A[i * 32 + kk] * (double)(float)B[n * 32 + kk];
            const double u = ldexp(1.0, (int)floor(log2(fabs(x) + 1e-300)) - 23);
            e32 += pow((D32[i * 32 + n] - x) / u, 2); n32++; ulp += u;
            if (i < 16 && n < 16) {
               double y = C[i * 32 + n]; for (int kk = 0; kk < 32; kk++) y += (double)(float)A[i * 32 + kk] * (double)(float)B[n * 32 + kk];
               const double u2 = ldexp(1.0, (int)floor(log2(fabs(y) + 1e-300)) - 23);
               e16 += pow((D16[i * 32 + n] - y) / u2, 2); n16++;
               double z = C[i * 32 + n]; for (int kk = 0; kk < 32; kk++) z += (double)(float)(__bf16)(float)A[i * 32 + kk] * (double)(float)(__bf16)(float)B[n * 32 + kk];
               const double u3 = ldexp(1.0, (int)floor(log2(fabs(z) + 1e-300)) - 23);
               eb += pow((Db[i * 32 + n] - z) / u3, 2);
            }
         }
      }
      printf("accumulator ~%g, operands ~%g: rms error in ulps of the result: f16 32x32x16 %.3f   f16 16x16x32 %.3f   bf16 16x16x32 %.3f   (round to nearest: 0.289)\n", cmag, pmag, sqrt(e32 / n32), sqrt(e16 / n16), sqrt(eb / n16));
   }
   return 0;
}
