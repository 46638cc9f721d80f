// gmm_exact.hip -- K1: diagonal-covariance GMM state log-likelihoods, bit-exact to the reference.
//
// Replaces IDOutP (HModel.c:5420-5431) + the mixture loop of ShStrP (HFB.c:949-960) /
// cSOutP (HRec.c:471-491):
//     sum = gConst; for i: xmm = x[i]-mean[i]; sum += xmm*xmm*ivar[i];   (float, in order, no FMA)
//     mixp = -0.5*sum;  x = LAdd(x, wt+mixp)   (float add; LAdd in double, HMath.c:1576; stored to float)
//
// MI355X mapping.  Lanes are FRAMES, the Gaussian is wave-uniform: one wave owns a tile of
// 64*FPL frames of one utterance, keeps those feature vectors in VGPRs (D*FPL registers) and
// walks over the tied states of its task.  Because the state/component/Gaussian indices are
// wave-uniform, the parameter table (gconst, then interleaved mean/ivar pairs, 16-byte aligned
// rows) is fetched with scalar loads (s_load_dwordx8/x16 through the scalar cache) and used as
// SGPR operands of v_sub/v_mul/v_mul/v_add: no LDS, no VGPR traffic for parameters, and every
// byte of the feature matrix is read from HBM once per task (coalesced 64*D*4-byte tile).
// Output is state-major (out[slot*ldo + t]) so that the 64 lanes store 256 contiguous bytes.
//
// Algorithmic work: M*(4*D+8) flop per (frame,state); VALU bound.
#include <hip/hip_runtime.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"

__device__ __forceinline__ double dev_ladd(double x, double y, double minLogExp)
{
   if (x < y) { double t = x; x = y; y = t; }
   double diff = y - x;
   if (diff < minLogExp) return (x < LSMALL) ? LZERO : x;
   return x + log(1.0 + exp(diff));
}

template <int D, int FPL>
__global__ __launch_bounds__(256) void k_score_exact(ScoreArgs a)
{
   const int lane = threadIdx.x & 63;
   const int task = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
   if (task >= a.nTasks) return;
   const ScoreTask tk = a.tasks[task];

   float x[FPL][D];
#pragma unroll
   for (int f = 0; f < FPL; f++) {
      int t = lane + 64 * f;
      if (t > tk.nFrames - 1) t = tk.nFrames - 1;
      const float *row = a.X + (size_t)(tk.frame0 + t) * D;
#pragma unroll
      for (int i = 0; i < D; i++) x[f][i] = row[i];
   }

   for (int k = 0; k < tk.nSlots; k++) {
      const int s = a.slotState[tk.slot0 + k];
      const int c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
      float acc[FPL];
      if (c1 - c0 == 1) {                       // single Gaussian: no weight, no LAdd (HFB.c:917-928)
         const float *P = a.gparam + (size_t)a.compGauss[c0] * a.PS;
         float sum[FPL];
#pragma unroll
         for (int f = 0; f < FPL; f++) sum[f] = P[0];
#pragma unroll
         for (int i = 0; i < D; i++) {
            const float mu = P[1 + 2 * i], iv = P[2 + 2 * i];
#pragma unroll
            for (int f = 0; f < FPL; f++) {
               float xmm = x[f][i] - mu;
               sum[f] += xmm * xmm * iv;
            }
         }
#pragma unroll
         for (int f = 0; f < FPL; f++) acc[f] = -0.5f * sum[f];
      } else {
#pragma unroll
         for (int f = 0; f < FPL; f++) acc[f] = (float)LZERO;
         for (int c = c0; c < c1; c++) {
            const float wt = a.compLogWt[c];
            if (wt > (float)LMINMIX) {          // wave-uniform branch
               const float *P = a.gparam + (size_t)a.compGauss[c] * a.PS;
               float sum[FPL];
#pragma unroll
               for (int f = 0; f < FPL; f++) sum[f] = P[0];
#pragma unroll
               for (int i = 0; i < D; i++) {
                  const float mu = P[1 + 2 * i], iv = P[2 + 2 * i];
#pragma unroll
                  for (int f = 0; f < FPL; f++) {
                     float xmm = x[f][i] - mu;
                     sum[f] += xmm * xmm * iv;
                  }
               }
#pragma unroll
               for (int f = 0; f < FPL; f++) {
                  float mixp = -0.5f * sum[f];
                  float y = wt + mixp;
                  acc[f] = (float)dev_ladd((double)acc[f], (double)y, a.minLogExp);
               }
            }
         }
      }
#pragma unroll
      for (int f = 0; f < FPL; f++) {
         int t = lane + 64 * f;
         if (t < tk.nFrames) a.out[tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + t] = acc[f];
      }
   }
}

// Any vector size: features are re-read from global memory per dimension (L1-resident rows).
__global__ __launch_bounds__(256) void k_score_exact_anyD(ScoreArgs a)
{
   const int lane = threadIdx.x & 63;
   const int task = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
   if (task >= a.nTasks) return;
   const ScoreTask tk = a.tasks[task];
   const int D = a.D;
   for (int f = 0; f < 2; f++) {
      int t = lane + 64 * f;
      const bool live = t < tk.nFrames;
      if (!live) t = tk.nFrames - 1;
      const float *row = a.X + (size_t)(tk.frame0 + t) * D;
      for (int k = 0; k < tk.nSlots; k++) {
         const int s = a.slotState[tk.slot0 + k];
         const int c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
         float acc = (float)LZERO;
         for (int c = c0; c < c1; c++) {
            const float wt = a.compLogWt[c];
            if (c1 - c0 > 1 && !(wt > (float)LMINMIX)) continue;
            const float *P = a.gparam + (size_t)a.compGauss[c] * a.PS;
            float sum = P[0];
            for (int i = 0; i < D; i++) {
               float xmm = row[i] - P[1 + 2 * i];
               sum += xmm * xmm * P[2 + 2 * i];
            }
            float mixp = -0.5f * sum;
            if (c1 - c0 == 1) acc = mixp;
            else {
               float y = wt + mixp;
               acc = (float)dev_ladd((double)acc, (double)y, a.minLogExp);
            }
         }
         if (live) a.out[tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + t] = acc;
      }
   }
}

int htkamd_launch_score_exact(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream)
{
   if (a.nTasks <= 0) return HTKAMD_OK;
   dim3 grid((a.nTasks + 3) / 4), block(256);
   switch (m->D) {
   case 39: hipLaunchKernelGGL((k_score_exact<39, 2>), grid, block, 0, stream, a); break;
   case 26: hipLaunchKernelGGL((k_score_exact<26, 2>), grid, block, 0, stream, a); break;
   case 13: hipLaunchKernelGGL((k_score_exact<13, 2>), grid, block, 0, stream, a); break;
   default: hipLaunchKernelGGL(k_score_exact_anyD, grid, block, 0, stream, a); break;
   }
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" int htkamd_outp_block(htkamd_model *m, const float *dX, int T, const int *dStates, int ns,
                                 float *dOut, int ldo, void *stream)
{
   if (!m || !dX || !dStates || !dOut || T < 0 || ns < 0 || ldo < T) {
      htkamd_set_error("outp_block: bad argument (T=%d ns=%d ldo=%d)", T, ns, ldo);
      return HTKAMD_EINVAL;
   }
   if (T == 0 || ns == 0) return HTKAMD_OK;
   hipStream_t s = (hipStream_t)stream;
   const int FR = SCORE_TILE_FRAMES, SL = SCORE_TASK_SLOTS;
   const int nTiles = (T + FR - 1) / FR, nChunks = (ns + SL - 1) / SL;
   const int nTasks = nTiles * nChunks;
   ScoreTask *h = (ScoreTask *)malloc(sizeof(ScoreTask) * (size_t)nTasks);
   int n = 0;
   for (int ti = 0; ti < nTiles; ti++)
      for (int ch = 0; ch < nChunks; ch++) {
         ScoreTask &tk = h[n++];
         tk.frame0 = ti * FR;
         tk.nFrames = (T - ti * FR < FR) ? T - ti * FR : FR;
         tk.slot0 = ch * SL;
         tk.nSlots = (ns - ch * SL < SL) ? ns - ch * SL : SL;
         tk.outSlot0 = ch * SL;
         tk.ldo = ldo;
         tk.outBase = (size_t)ti * FR;
      }
   ScoreTask *d = nullptr;
   HIPCHECK(hipMalloc((void **)&d, sizeof(ScoreTask) * (size_t)nTasks));
   HIPCHECK(hipMemcpyAsync(d, h, sizeof(ScoreTask) * (size_t)nTasks, hipMemcpyHostToDevice, s));
   ScoreArgs a;
   a.tasks = d; a.nTasks = nTasks; a.X = dX; a.slotState = dStates; a.out = dOut;
   a.stateCompOff = m->d_stateCompOff; a.compGauss = m->d_compGauss; a.compLogWt = m->d_compLogWt;
   a.gparam = m->d_gparam; a.PS = m->PS; a.D = m->D; a.minLogExp = m->minLogExp;
   int rc = htkamd_launch_score_exact(m, a, s);
   hipError_t e = hipStreamSynchronize(s);      // the task table is freed below
   free(h);
   (void)hipFree(d);
   if (rc) return rc;
   HIPCHECK(e);
   return HTKAMD_OK;
}
