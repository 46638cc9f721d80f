// update.hip -- UpdateModels on the device: re-estimation from the summed accumulator vector and the refresh of every table the
// scoring / recursion kernels read, without the accumulators or the parameters crossing PCIe.
//
// Same arithmetic as htk_amd/host/update.c (= HERest.c MLUpdateModels :1262, UpdateTrans :795, UpdateWeights :897 + FloorMixes
// :819, UpdateVars :1045, UpdateMeans :974, FixGConsts HModel.c:5688): every fp64 sum is rounded to float ONCE and the reference's
// float expressions follow (no FMA contraction; correctly rounded float division).  The reference walks the models in scan order
// and lets the first model that qualifies (>= minEgs examples) update a shared structure; since a structure is updated at most once,
// that is "update exactly the structures some qualifying model reaches", which is order-free:
//   k_upd_mark    thread = physical HMM: example count against minEgs; marks its transition matrix and its tied states
//   k_upd_trans   thread = transition matrix (marked): a_ij = tr/occ -> log
//   k_upd_state   thread = tied state: [single-process float round trip of the weights]; marked: c_m/occ, MINMIX cut, floor; then the
//                 log weights the kernels use, and the state's Gaussians whose new weight exceeds MINMIX are marked
//   k_upd_gauss_elem  thread = (Gaussian, dimension): [round trip of the variance]; marked: variance (from the OLD mean's statistics,
//                 as UpdateVars runs before UpdateMeans), mean; then 1/variance and the interleaved (mean, 1/var) scoring row
//   k_upd_gconst  thread = Gaussian: gConst (float sum in dimension order)
//   k_upd_mfma    thread = (fragment tile, component column): the A-operand table of the matrix-core scoring kernel
// 80 000 Gaussians x 39 dimensions: ~60 MB read, ~60 MB written, a fraction of a millisecond -- against 52 MB D2H, a single host
// thread over 80 k Gaussians and ~80 MB H2D on the host path (which stays: htkamd_model_update).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "internal.h"
#include "hipcheck.h"

struct UpdArgs {
   int D, S, C, G, nT, H, PS, maxM;
   const int *stateCompOff, *compGauss, *transN, *transOff, *trOccOff, *hmmTrans, *hmmStateOff, *hmmState;
   float *mean, *var, *gconst, *compWeight, *transP;        // parameters (DIAGC variances, linear weights, log transitions)
   float *ivar, *gparam, *compLogWt;                        // derived tables
   const unsigned char *rawLogWt;                           // HTKAMD_COMPAT_SHARED_LOGWT: components whose weight ConvLogWt skipped (internal.h), or NULL
   const double *acc;
   htkamd_accs_layout lay;
   int minEgs, uFlags, singleProcess, rowNormalise, hasVarFloor;
   float minVar, mixWeightFloor;
   double logTpi;                                           // log(2 pi) as the host's libm gives it
   const float *varFloor;
   unsigned char *qualT, *qualS, *qualG, *anyS, *anyG;      // marked by a qualifying model / used by any model
   unsigned char *flooredG;                                 // a variance element of the Gaussian was floored
   int *stats;                                              // htkamd_update_stats fields in order + [6] weights above 1.001
   int *blkStats;                                           // k_upd_gauss_fused: [blocks][3] counters of the blocks (in logVar's place)
   float *logVar;                                           // [G*D] log of the new variances (k_upd_gauss_elem) for k_upd_gconst's sum; NULL: tied sets
   // sets with tied mean / variance vectors (~u ~v; htkamd_model_set_sharing), else tied == 0
   int tied;
   const int *meanLeader, *varLeader, *varGroupSize, *muMemOff, *vaMemOff, *muMem, *vaMem, *scanPos;
   int *firstMu, *firstVa;                                  // scan position of the first qualifying model whose mixture reaches the vector
   const int *dimStream, *gaussStream;                      // several streams: stream of a dimension / of a Gaussian (NULL: one stream)
};
// a dimension outside the Gaussian's stream: never re-estimated, 1/variance 0 (internal.h)
#define OUTSIDE(g, k) (a.dimStream != nullptr && a.dimStream[k] != a.gaussStream[g])

#define ACCF(off, idx) ((float)a.acc[(off) + (size_t)(idx)])

__global__ void k_upd_mark(UpdArgs a)
{
   const int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (h >= a.H) return;
   const long long n = llround(a.acc[a.lay.nEgs + h]);
   const int *hs = a.hmmState + a.hmmStateOff[h];
   const int ns = a.hmmStateOff[h + 1] - a.hmmStateOff[h];
   for (int j = 0; j < ns; j++) a.anyS[hs[j]] = 1;
   if (n < a.minEgs) atomicAdd(a.stats + 2, 1);
   if (!(n >= a.minEgs && n > 0)) return;
   a.qualT[a.hmmTrans[h]] = 1;
   for (int j = 0; j < ns; j++) a.qualS[hs[j]] = 1;
}

__device__ __forceinline__ void upd_trans_one(const UpdArgs &a, const int ti)
{
   if (ti >= a.nT || !a.qualT[ti] || !(a.uFlags & HTKAMD_UPTRANS)) return;
   const int N = a.transN[ti];
   float *tp = a.transP + a.transOff[ti];
   for (int i = 1; i < N; i++) {
      const float occi = ACCF(a.lay.trOcc, a.trOccOff[ti] + i - 1);
      if (occi > 0.0f && a.rowNormalise) {                   // RestTransP (HRest.c:1015-1039)
         float sum = 0.0f;
         for (int j = 2; j <= N; j++) sum += ACCF(a.lay.tr, a.transOff[ti] + (i - 1) * N + (j - 1)) / occi;
         for (int j = 2; j <= N; j++) {
            const float x = (ACCF(a.lay.tr, a.transOff[ti] + (i - 1) * N + (j - 1)) / occi) / sum;
            tp[(i - 1) * N + (j - 1)] = ((double)x < MINLARG) ? (float)LZERO : (float)log((double)x);
         }
      } else if (occi > 0.0f) {
         for (int j = 2; j <= N; j++) {
            const float x = ACCF(a.lay.tr, a.transOff[ti] + (i - 1) * N + (j - 1)) / occi;
            tp[(i - 1) * N + (j - 1)] = ((double)x > MINLARG) ? (float)log((double)x) : (float)LZERO;
         }
      } else atomicAdd(a.stats + 3, 1);
   }
}

__global__ void k_upd_trans(UpdArgs a) { upd_trans_one(a, blockIdx.x * blockDim.x + threadIdx.x); }

__device__ __forceinline__ float mix_log_weight(float w) { return ((double)w < MINMIX) ? (float)LZERO : (float)log((double)w); }

__global__ void k_upd_state(UpdArgs a)
{
   const int s = blockIdx.x * blockDim.x + threadIdx.x;
   if (s >= a.S) return;
   const int c0 = a.stateCompOff[s], M = a.stateCompOff[s + 1] - c0;
   float *wgt = a.compWeight + c0;
   if (a.singleProcess && a.anyS[s])                         // ConvLogWt before the pass, ConvExpWt after it (HERest.c:1336-1339)
      for (int k = 0; k < M; k++) if (!(a.rawLogWt && a.rawLogWt[c0 + k])) wgt[k] = (float)exp((double)mix_log_weight(wgt[k]));
   if (a.qualS[s] && a.maxM > 1 && (a.uFlags & HTKAMD_UPMIXES)) {
      const float occi = ACCF(a.lay.wtOcc, s);
      if (occi > 0.0f) {
         for (int k = 0; k < M; k++) {
            float x = ACCF(a.lay.wt, c0 + k) / occi;
            if ((double)x > 1.001) atomicAdd(a.stats + 6, 1);                 // HError 2393 in the reference
            if (x > 1.0f) x = 1.0f;
            wgt[k] = ((double)x > MINMIX) ? x : 0.0f;
         }
         if (a.mixWeightFloor > 0.0f) {                      // FloorMixes
            float sum = 0.0f, fsum = 0.0f;
            const float floor = a.mixWeightFloor;
            for (int k = 0; k < M; k++) {
               if (wgt[k] > floor) sum += wgt[k];
               else { fsum += floor; wgt[k] = floor; }
            }
            if (fsum != 0.0f && sum != 0.0f) {
               const float scale = (float)((1.0 - (double)fsum) / (double)sum);
               for (int k = 0; k < M; k++) if (wgt[k] > floor) wgt[k] *= scale;
            }
         }
      } else atomicAdd(a.stats + 4, 1);
   }
   for (int k = 0; k < M; k++) {
      a.compLogWt[c0 + k] = (a.rawLogWt && a.rawLogWt[c0 + k]) ? wgt[k] : mix_log_weight(wgt[k]);
      const int g = a.compGauss[c0 + k];
      if (a.anyS[s]) a.anyG[g] = 1;
      if (a.qualS[s] && (double)wgt[k] > MINMIX) a.qualG[g] = 1;
   }
}

// The same with one thread per mixture component, GW (a power of two >= the largest mixture) lanes to a state: the double exp() / log() of
// the conversions run side by side; the two float sums of FloorMixes are taken by every lane over the group's weights in the reference's
// order.  5 000 threads of k_upd_state were 23 us of latency at 5k x 16.
template <int GW>
__global__ __launch_bounds__(256) void k_upd_state_w(UpdArgs a, const int nbState)
{
   if ((int)blockIdx.x >= nbState) { upd_trans_one(a, (blockIdx.x - nbState) * blockDim.x + threadIdx.x); return; }      // the transitions ride along
   const int tid = blockIdx.x * blockDim.x + threadIdx.x;
   const int s = tid / GW, k = tid % GW;
   const bool sIn = s < a.S;
   const int sc = sIn ? s : 0;
   const int c0 = a.stateCompOff[sc], M = a.stateCompOff[sc + 1] - c0;
   const bool in = sIn && k < M;
   const int c = in ? c0 + k : c0;
   const bool raw = a.rawLogWt && a.rawLogWt[c];
   const unsigned char anyS = a.anyS[sc], qualS = a.qualS[sc];
   float w = a.compWeight[c];
   if (a.singleProcess && anyS && !raw) w = (float)exp((double)mix_log_weight(w));
   if (qualS && a.maxM > 1 && (a.uFlags & HTKAMD_UPMIXES)) {
      const float occi = ACCF(a.lay.wtOcc, sc);
      if (occi > 0.0f) {
         float x = ACCF(a.lay.wt, c) / occi;
         if (in && (double)x > 1.001) atomicAdd(a.stats + 6, 1);
         if (x > 1.0f) x = 1.0f;
         w = ((double)x > MINMIX) ? x : 0.0f;
         if (a.mixWeightFloor > 0.0f) {                      // FloorMixes
            const float floor = a.mixWeightFloor;
            float sum = 0.0f, fsum = 0.0f;
            for (int j = 0; j < GW; j++) {
               const float wj = __shfl(w, j, GW);
               if (j < M) { if (wj > floor) sum += wj; else fsum += floor; }
            }
            if (!(w > floor)) w = floor;
            else if (fsum != 0.0f && sum != 0.0f) w *= (float)((1.0 - (double)fsum) / (double)sum);
         }
      } else if (sIn && k == 0) atomicAdd(a.stats + 4, 1);
   }
   if (!in) return;
   a.compWeight[c] = w;
   a.compLogWt[c] = raw ? w : mix_log_weight(w);
   const int g = a.compGauss[c];
   if (anyS) a.anyG[g] = 1;
   if (qualS && (double)w > MINMIX) a.qualG[g] = 1;
}

// Tied vectors.  The reference hangs ONE accumulator on a shared vector and lets the first mixture that reaches it (models in scan
// order; in a model all variances, then all means: HERest.c:1262-1321) update it; a variance loses its mean-shift term when its mean
// was moved by an EARLIER model (HERest.c:1080 `shared`).  "First" is a minimum over scan positions, which needs no order:
__global__ void k_upd_first(UpdArgs a)
{
   const int h = blockIdx.x * blockDim.x + threadIdx.x;
   if (h >= a.H) return;
   const long long n = llround(a.acc[a.lay.nEgs + h]);
   if (!(n >= a.minEgs && n > 0)) return;
   const int pos = a.scanPos ? a.scanPos[h] : h;
   const int *hs = a.hmmState + a.hmmStateOff[h];
   const int ns = a.hmmStateOff[h + 1] - a.hmmStateOff[h];
   for (int j = 0; j < ns; j++)
      for (int c = a.stateCompOff[hs[j]]; c < a.stateCompOff[hs[j] + 1]; c++)
         if ((double)a.compWeight[c] > MINMIX) {
            const int g = a.compGauss[c];
            atomicMin(a.firstMu + a.meanLeader[g], pos);
            atomicMin(a.firstVa + a.varLeader[g], pos);
         }
}

// the group's statistic: the leader's plus its members' in ascending order, as htkamd_update_models pools them (fp64)
__device__ __forceinline__ double pooled(const double *acc, size_t off, int leader, int stride, int k, const int *memOff, const int *mem)
{
   double v = acc[off + (size_t)leader * stride + k];
   for (int i = memOff[leader]; i < memOff[leader + 1]; i++) v += acc[off + (size_t)mem[i] * stride + k];
   return v;
}

// Every member of a group computes the group's new vector from the pooled statistics and its OWN (equal) copy of the old one, so no
// copy pass is needed and no thread reads a slot another one writes; the counts are taken by the leader's threads.
__global__ void k_upd_gauss_elem_tied(UpdArgs a)
{
   const size_t idx0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   const int D = a.D;
   const bool live = idx0 < (size_t)a.G * D;
   const unsigned int idx = live ? (unsigned int)idx0 : 0u;
   const int g = (int)(idx / (unsigned int)D), k = (int)(idx - (unsigned int)g * (unsigned int)D);
   const int gv = a.varLeader[g], gm = a.meanLeader[g];
   float v = a.var[idx], mu = a.mean[idx];
   bool floored = false;
   const bool outside = OUTSIDE(g, k);
   if (a.singleProcess && a.anyG[gv] && !outside) {
      float iv;
      if (v > 1E+30f) v = 1E+30f;
      if (v < 1E-30f) v = 1E-30f;
      iv = 1 / v;
      if (iv > 1E+30f) iv = 1E+30f;
      if (iv < 1E-30f) iv = 1E-30f;
      v = 1 / iv;
   }
   const int posVa = a.firstVa[gv], posMu = a.firstMu[gm];
   if ((a.uFlags & HTKAMD_UPVARS) && posVa != 0x7f7f7f7f && !outside) {
      const float occim = (float)pooled(a.acc, a.lay.vaOcc, gv, 1, 0, a.vaMemOff, a.vaMem);
      if (occim > 0.0f) {
         // the mean of the mixture that reached this variance first: for a private variance its own Gaussian's
         const float muOcc = (float)pooled(a.acc, a.lay.muOcc, gm, 1, 0, a.muMemOff, a.muMem);
         const bool shared = (a.uFlags & HTKAMD_UPMEANS) == 0 || posMu < posVa || muOcc <= 0.0f || a.varGroupSize[gv] > 1;
         const float muDiffk = shared ? 0.0f : (float)pooled(a.acc, a.lay.mu, gm, D, k, a.muMemOff, a.muMem) / muOcc;
         float x = (float)pooled(a.acc, a.lay.va, gv, D, k, a.vaMemOff, a.vaMem) / occim - muDiffk * muDiffk;
         const float fl = a.hasVarFloor ? a.varFloor[k] : a.minVar;
         if (x < fl) { x = fl; floored = live && g == gv; }
         v = x;
      } else if (k == 0 && live && g == gv) atomicAdd(a.stats + 5, 1);
   }
   if ((a.uFlags & HTKAMD_UPMEANS) && posMu != 0x7f7f7f7f && !outside) {
      const float muOcc = (float)pooled(a.acc, a.lay.muOcc, gm, 1, 0, a.muMemOff, a.muMem);
      if (muOcc > 0.0f) mu += (float)pooled(a.acc, a.lay.mu, gm, D, k, a.muMemOff, a.muMem) / muOcc;
   }
   {
      const unsigned long long fb = __ballot(floored);
      if (fb && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)fb) - 1)) atomicAdd(a.stats + 0, __popcll(fb));
      if (floored) a.flooredG[g] = 1;
   }
   if (!live) return;
   a.var[idx] = v; a.mean[idx] = mu;
   float c = v;
   if (c > 1E+30f) c = 1E+30f;
   if (c < 1E-30f) c = 1E-30f;
   const float r = outside ? 0.0f : 1 / c;
   a.ivar[idx] = r;
   float *gp = a.gparam + (size_t)g * a.PS;
   gp[2 * k] = mu; gp[2 * k + 1] = r;
}

// Variances and means, one thread per (Gaussian, dimension): every array is walked in storage order (coalesced), the per-Gaussian
// quantities (marks, occupancies) are broadcast loads.
__global__ __launch_bounds__(256) void k_upd_gauss_elem(UpdArgs a)
{
   // four consecutive elements per thread: the parameter arrays move as 16-byte words (the kernel was ~1 TB/s of 4- and 8-byte accesses)
   const unsigned int nEl = (unsigned int)a.G * (unsigned int)a.D;      // < 2^31 (checked by the launcher)
   const unsigned int i0 = ((unsigned int)blockIdx.x * blockDim.x + threadIdx.x) * 4u;
   const int D = a.D;
   const bool any4 = i0 < nEl;
   const bool full = i0 + 3u < nEl;
   float v[4], mu[4], r[4];
   double dMu[4], dVa[4];
   int g[4], k[4];
   bool live[4], floored[4];
   if (full) {
      const float4 v4 = *(const float4 *)(a.var + i0), m4 = *(const float4 *)(a.mean + i0);
      v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w; mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
   }
#pragma unroll
   for (int j = 0; j < 4; j++) {
      const unsigned int idx = (i0 + j < nEl) ? i0 + j : 0u;
      live[j] = i0 + j < nEl;
      g[j] = (int)(idx / (unsigned int)D); k[j] = (int)(idx - (unsigned int)g[j] * (unsigned int)D);
      if (!full) { v[j] = a.var[idx]; mu[j] = a.mean[idx]; }
      dMu[j] = a.acc[a.lay.mu + idx]; dVa[j] = a.acc[a.lay.va + idx];
      floored[j] = false;
   }
   // per-Gaussian quantities: the four elements lie in one Gaussian or in two
   const double oMu0 = a.acc[a.lay.muOcc + g[0]], oVa0 = a.acc[a.lay.vaOcc + g[0]], oMu3 = a.acc[a.lay.muOcc + g[3]], oVa3 = a.acc[a.lay.vaOcc + g[3]];
   const unsigned char q0 = a.qualG[g[0]], q3 = a.qualG[g[3]], an0 = a.anyG[g[0]], an3 = a.anyG[g[3]];
#pragma unroll
   for (int j = 0; j < 4; j++) {
      const bool lo = g[j] == g[0];                           // D >= 4 is not assumed: a third Gaussian in between is read on its own
      const bool hi = g[j] == g[3];
      const double dMuOcc = lo ? oMu0 : hi ? oMu3 : a.acc[a.lay.muOcc + g[j]], dVaOcc = lo ? oVa0 : hi ? oVa3 : a.acc[a.lay.vaOcc + g[j]];
      const unsigned char qual = lo ? q0 : hi ? q3 : a.qualG[g[j]], any = lo ? an0 : hi ? an3 : a.anyG[g[j]];
      const bool outside = OUTSIDE(g[j], k[j]);
      float vv = v[j], mm = mu[j];
      if (a.singleProcess && any && !outside) {              // ConvDiagC before the pass, ForceDiagC after it
         float iv;
         if (vv > 1E+30f) vv = 1E+30f;
         if (vv < 1E-30f) vv = 1E-30f;
         iv = 1 / vv;
         if (iv > 1E+30f) iv = 1E+30f;
         if (iv < 1E-30f) iv = 1E-30f;
         vv = 1 / iv;
      }
      if (qual && !outside) {
         const float muOcc = (float)dMuOcc;
         if (a.uFlags & HTKAMD_UPVARS) {
            const float occim = (float)dVaOcc;
            if (occim > 0.0f) {
               const bool shared = (a.uFlags & HTKAMD_UPMEANS) == 0 || muOcc <= 0.0f;
               const float muDiffk = shared ? 0.0f : (float)dMu[j] / muOcc;
               float x = (float)dVa[j] / occim - muDiffk * muDiffk;
               const float fl = a.hasVarFloor ? a.varFloor[k[j]] : a.minVar;
               if (x < fl) { x = fl; floored[j] = live[j]; }
               vv = x;
            } else if (k[j] == 0 && live[j]) atomicAdd(a.stats + 5, 1);
         }
         if ((a.uFlags & HTKAMD_UPMEANS) && muOcc > 0.0f) mm += (float)dMu[j] / muOcc;
      }
      v[j] = vv; mu[j] = mm;
      // derived tables: ConvDiagC (HUtil.c:413) and the interleaved row of the exact scoring kernel
      float c = vv;
      if (c > 1E+30f) c = 1E+30f;
      if (c < 1E-30f) c = 1E-30f;
      r[j] = outside ? 0.0f : 1 / c;
   }
   {  // floored elements: one atomic per wavefront
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) { cnt += __popcll(__ballot(floored[j])); if (floored[j]) a.flooredG[g[j]] = 1; }
      if (cnt && (threadIdx.x & 63) == 0) atomicAdd(a.stats + 0, cnt);
   }
   if (!any4) return;
   // the terms of gConst (FixDiagGConst HModel.c:5641: log evaluated in double, rounded to float) where this kernel has ALU time to spare
   float z[4];
#pragma unroll
   for (int j = 0; j < 4; j++) z[j] = ((double)v[j] <= MINLARG) ? (float)LZERO : (float)log((double)v[j]);
   if (full) {
      *(float4 *)(a.var + i0) = make_float4(v[0], v[1], v[2], v[3]);
      *(float4 *)(a.mean + i0) = make_float4(mu[0], mu[1], mu[2], mu[3]);
      *(float4 *)(a.ivar + i0) = make_float4(r[0], r[1], r[2], r[3]);
      *(float4 *)(a.logVar + i0) = make_float4(z[0], z[1], z[2], z[3]);
   }
#pragma unroll
   for (int j = 0; j < 4; j++) {
      if (!live[j]) continue;
      if (!full) { a.var[i0 + j] = v[j]; a.mean[i0 + j] = mu[j]; a.ivar[i0 + j] = r[j]; a.logVar[i0 + j] = z[j]; }
      *(float2 *)(a.gparam + (size_t)g[j] * a.PS + 2 * k[j]) = make_float2(mu[j], r[j]);
   }
}

// Variances, means AND gConst in one pass (round 5): a workgroup owns UPD_GPB whole Gaussians -- their elements as 16-byte words in storage
// order (coalesced, as k_upd_gauss_elem), the logs of the new variances into LDS, then one thread per Gaussian sums its row in the
// reference's order (FixDiagGConst HModel.c:5641) and closes the Gaussian's row of the scoring table.  Saves the separate gConst launch
// and the write + read of the log-variance array (25 MB at 5k x 16).  Sets without shared vectors and with G D divisible by 4.
#define UPD_GPB 64
template <int NIT>                                         // 16-byte words per thread: ceil(UPD_GPB D / 1024)
__global__ __launch_bounds__(256) void k_upd_gauss_fused(UpdArgs a)
{
   // [UPD_GPB][PS] the block's rows of the exact kernel's table (mean, inverse variance interleaved; gConst; padding), [UPD_GPB][D] log
   // variances, [UPD_GPB] floored flags.  The table rows leave as whole 16-byte words: written pair by pair from the element loop they were
   // 8-byte stores 32 bytes apart, and the kernel's HBM writes were twice its data (WRITE_SIZE, profiles/README.md r05b).
   extern __shared__ float updLds[];
   const int D = a.D, PS = a.PS;
   const int g0 = blockIdx.x * UPD_GPB;
   const int nG = (a.G - g0 < UPD_GPB) ? a.G - g0 : UPD_GPB;
   float *gpRows = updLds, *lvRows = updLds + (size_t)UPD_GPB * PS;
   int *flRow = (int *)(lvRows + (size_t)UPD_GPB * D);
   // the block's counters (floored elements, floored Gaussians, Gaussians without variance statistics): summed by k_upd_export.  One atomic
   // per wavefront on the set's counter was 5 000 updates of one address, 7 ns each: 35 of the kernel's 76 us
   __shared__ int blkCnt[4];
   if (threadIdx.x < 4) blkCnt[threadIdx.x] = 0;
   for (int i = threadIdx.x; i < UPD_GPB; i += blockDim.x) flRow[i] = 0;
   for (int i = threadIdx.x; i < nG; i += blockDim.x)
      for (int k = 2 * D + 1; k < PS; k++) gpRows[i * PS + k] = 0.0f;                       // (the padding of a row is zeros: gparam_refresh, model.hip)
   __syncthreads();
   const unsigned int e0 = (unsigned int)g0 * (unsigned int)D, nEl = (unsigned int)nG * (unsigned int)D;      // the block's elements: e0 .. e0 + nEl
   const bool accPairs = ((a.lay.mu | a.lay.va) & 1) == 0 && (((size_t)a.acc & 15) == 0);   // the statistics as 16-byte words
   int nFloored = 0;
   // (UPD_HOIST=1: every load of the thread before the first element is worked on -- measured slower, 48 us against 39)
   float4 vAll[NIT], mAll[NIT];
   double dMuAll[NIT][4], dVaAll[NIT][4];
   auto load_it = [&](const int it) {
      const unsigned int c = (threadIdx.x + it * 256u) * 4u, i0 = e0 + c;
      vAll[it] = mAll[it] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
      for (int j = 0; j < 4; j++) dMuAll[it][j] = dVaAll[it][j] = 0.0;
      if (c + 3u < nEl) {
         vAll[it] = *(const float4 *)(a.var + i0); mAll[it] = *(const float4 *)(a.mean + i0);
         if (accPairs) {
            const double2 m0 = *(const double2 *)(a.acc + a.lay.mu + i0), m1 = *(const double2 *)(a.acc + a.lay.mu + i0 + 2);
            const double2 q0 = *(const double2 *)(a.acc + a.lay.va + i0), q1 = *(const double2 *)(a.acc + a.lay.va + i0 + 2);
            dMuAll[it][0] = m0.x; dMuAll[it][1] = m0.y; dMuAll[it][2] = m1.x; dMuAll[it][3] = m1.y;
            dVaAll[it][0] = q0.x; dVaAll[it][1] = q0.y; dVaAll[it][2] = q1.x; dVaAll[it][3] = q1.y;
         } else {
#pragma unroll
            for (int j = 0; j < 4; j++) { dMuAll[it][j] = a.acc[a.lay.mu + i0 + j]; dVaAll[it][j] = a.acc[a.lay.va + i0 + j]; }
         }
      } else if (c < nEl) {                                  // the set's last word, cut short
         float tv[4] = {1.0f, 1.0f, 1.0f, 1.0f}, tm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
         for (int j = 0; j < 4; j++)
            if (c + j < nEl) { tv[j] = a.var[i0 + j]; tm[j] = a.mean[i0 + j]; dMuAll[it][j] = a.acc[a.lay.mu + i0 + j]; dVaAll[it][j] = a.acc[a.lay.va + i0 + j]; }
         vAll[it] = make_float4(tv[0], tv[1], tv[2], tv[3]); mAll[it] = make_float4(tm[0], tm[1], tm[2], tm[3]);
      }
   };
#ifndef UPD_HOIST
#define UPD_HOIST 0
#endif
#if UPD_HOIST
#pragma unroll
   for (int it = 0; it < NIT; it++) load_it(it);
#endif
#pragma unroll
   for (int it = 0; it < NIT; it++) {
      const unsigned int c = (threadIdx.x + it * 256u) * 4u;
      if (c >= nEl) break;
#if !UPD_HOIST
      load_it(it);
#endif
      const unsigned int i0 = e0 + c;                       // a multiple of 4 (UPD_GPB D is)
      const bool full = c + 3u < nEl;
      float v[4] = {vAll[it].x, vAll[it].y, vAll[it].z, vAll[it].w}, mu[4] = {mAll[it].x, mAll[it].y, mAll[it].z, mAll[it].w}, r[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
         const bool live = c + j < nEl;
         const unsigned int cl = live ? c + j : 0u;
         const unsigned int idx = e0 + cl;
         const int gl = (int)(cl / (unsigned int)D), k = (int)(cl - (unsigned int)gl * (unsigned int)D), g = g0 + gl;
         const double dMu = dMuAll[it][j], dVa = dVaAll[it][j];
         const double dMuOcc = a.acc[a.lay.muOcc + g], dVaOcc = a.acc[a.lay.vaOcc + g];
         const unsigned char qual = a.qualG[g], any = a.anyG[g];
         const bool outside = OUTSIDE(g, k);
         float vv = v[j], mm = mu[j];
         bool floored = false;
         if (a.singleProcess && any && !outside) {              // ConvDiagC before the pass, ForceDiagC after it
            float iv;
            if (vv > 1E+30f) vv = 1E+30f;
            if (vv < 1E-30f) vv = 1E-30f;
            iv = 1 / vv;
            if (iv > 1E+30f) iv = 1E+30f;
            if (iv < 1E-30f) iv = 1E-30f;
            vv = 1 / iv;
         }
         if (qual && !outside) {
            const float muOcc = (float)dMuOcc;
            if (a.uFlags & HTKAMD_UPVARS) {
               const float occim = (float)dVaOcc;
               if (occim > 0.0f) {
                  const bool shared = (a.uFlags & HTKAMD_UPMEANS) == 0 || muOcc <= 0.0f;
                  const float muDiffk = shared ? 0.0f : (float)dMu / muOcc;
                  float x = (float)dVa / occim - muDiffk * muDiffk;
                  const float fl = a.hasVarFloor ? a.varFloor[k] : a.minVar;
                  if (x < fl) { x = fl; floored = live; }
                  vv = x;
               } else if (k == 0 && live) atomicAdd(blkCnt + 2, 1);
            }
            if ((a.uFlags & HTKAMD_UPMEANS) && muOcc > 0.0f) mm += (float)dMu / muOcc;
         }
         v[j] = vv; mu[j] = mm;
         float cc = vv;
         if (cc > 1E+30f) cc = 1E+30f;
         if (cc < 1E-30f) cc = 1E-30f;
         r[j] = outside ? 0.0f : 1 / cc;
         const float z = ((double)vv <= MINLARG) ? (float)LZERO : (float)log((double)vv);
         if (floored) { nFloored++; flRow[gl] = 1; }
         if (live) {
            lvRows[cl] = z;
            if (!full) { a.var[idx] = vv; a.mean[idx] = mm; a.ivar[idx] = r[j]; }
            *(float2 *)(gpRows + (size_t)gl * PS + 2 * k) = make_float2(mm, r[j]);
         }
      }
      if (full) {
         *(float4 *)(a.var + i0) = make_float4(v[0], v[1], v[2], v[3]);
         *(float4 *)(a.mean + i0) = make_float4(mu[0], mu[1], mu[2], mu[3]);
         *(float4 *)(a.ivar + i0) = make_float4(r[0], r[1], r[2], r[3]);
      }
   }
   {  // floored elements: one atomic per wavefront
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) nFloored += __shfl_xor(nFloored, o);
      if (nFloored && (threadIdx.x & 63) == 0) atomicAdd(blkCnt + 0, nFloored);
   }
   __syncthreads();
   // ---- gConst of the block's Gaussians, one thread each
   const int t = threadIdx.x;
   bool fl = false;
   if (t < nG) {
      const int g = g0 + t;
      fl = flRow[t] != 0;
      if (fl) a.flooredG[g] = 1;
      float gc = a.gconst[g];
      if (a.qualG[g] && (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS))) {
         int n = D;
         if (a.dimStream) { n = 0; for (int k = 0; k < D; k++) if (!OUTSIDE(g, k)) n++; }
         float sum = (float)((double)n * a.logTpi);
         const float *lv = lvRows + (size_t)t * D;
         for (int k = 0; k < D; k++) {
            if (OUTSIDE(g, k)) continue;
            sum += lv[k];
         }
         a.gconst[g] = gc = sum;
      }
      gpRows[(size_t)t * PS + 2 * D] = gc;
   }
   if (t < 64) {                                             // (the threads of the Gaussians are wavefront 0: UPD_GPB = 64)
      const unsigned long long fb = __ballot(fl);
      if (t == 0) { a.blkStats[3 * blockIdx.x + 0] = blkCnt[0]; a.blkStats[3 * blockIdx.x + 1] = __popcll(fb); a.blkStats[3 * blockIdx.x + 2] = blkCnt[2]; }
   }
   __syncthreads();
   // ---- the rows of the exact kernel's table, in storage order
   float4 *dst = (float4 *)(a.gparam + (size_t)g0 * PS);     // (PS is a multiple of 4 and the table 16-byte aligned)
   const float4 *src = (const float4 *)gpRows;
   for (int i = threadIdx.x; i < nG * (PS >> 2); i += blockDim.x) dst[i] = src[i];
}

// gConst, one thread per Gaussian: the float sum over the dimensions runs in the reference's order (FixDiagGConst HModel.c:5641)
__global__ __launch_bounds__(128) void k_upd_gconst(UpdArgs a)
{
   // the block's rows of log variances through LDS: read in storage order, summed by one thread per Gaussian in the reference's order
   extern __shared__ float lvRows[];                        // [128][D] when a.logVar is there
   const int g0 = blockIdx.x * blockDim.x, g = g0 + threadIdx.x;
   const int D = a.D;
   if (a.logVar) {
      const int nG = (a.G - g0 < (int)blockDim.x) ? a.G - g0 : (int)blockDim.x;
      const float *src = a.logVar + (size_t)g0 * D;
      for (int i = threadIdx.x; i < nG * D; i += blockDim.x) lvRows[i] = src[i];
      __syncthreads();
   }
   if (g >= a.G) return;
   int flooredCnt = a.flooredG[g] ? 1 : 0;
   {  // one atomic per wavefront
      const unsigned long long fb = __ballot(flooredCnt != 0);
      if (fb && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)fb) - 1)) atomicAdd(a.stats + 1, __popcll(fb));
   }
   if (a.qualG[g] && (a.uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS))) {
      const float *var = a.var + (size_t)g * D;
      int n = D;
      if (a.dimStream) { n = 0; for (int k = 0; k < D; k++) if (!OUTSIDE(g, k)) n++; }
      float sum = (float)((double)n * a.logTpi);
      const float *lv = a.logVar ? lvRows + (size_t)threadIdx.x * D : nullptr;
      for (int k = 0; k < D; k++) {
         if (OUTSIDE(g, k)) continue;
         const float z = lv ? lv[k] : (((double)var[k] <= MINLARG) ? (float)LZERO : (float)log((double)var[k]));
         sum += z;
      }
      a.gconst[g] = sum;
   }
   a.gparam[(size_t)g * a.PS + 2 * D] = a.gconst[g];
}

// A-operand fragment table of gmm_mfma.hip; layout and arithmetic as mfma_refresh() in model.hip
struct MfmaTabArgs {
   int D, NS, S;
   const int *stateCompOff, *stateTileOff, *compGauss;
   const float *mean, *ivar, *gconst, *compLogWt;
   float *tab;
};

__global__ void k_upd_mfma(MfmaTabArgs a, int nTiles)
{
   const int idx = blockIdx.x * blockDim.x + threadIdx.x;
   if (idx >= nTiles * 16) return;
   const int t = idx >> 4, col = idx & 15;
   // state of tile t: binary search in stateTileOff
   int lo = 0, hi = a.S - 1;
   while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.stateTileOff[mid] <= t) lo = mid; else hi = mid - 1; }
   const int s = lo, c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
   const int c = c0 + 16 * (t - a.stateTileOff[s]) + col;
   const int NS = a.NS, D = a.D;
   float *T = a.tab + (size_t)t * (NS + 8) * 64;
   float *ciRow = T + (size_t)(NS + (col & 3)) * 64 + (col >> 2) * 16, *endRow = ciRow + 4 * 64;
   const bool live = c < c1 && (c1 - c0 == 1 || a.compLogWt[c] > (float)LMINMIX);
   if (!live) {
      for (int j = 0; j < 16; j++) { ciRow[j] = 0.0f; endRow[j] = -1.0e30f; }
      for (int st = 0; st < NS; st++) for (int kq = 0; kq < 4; kq++) T[(size_t)st * 64 + kq * 16 + col] = 0.0f;
      return;
   }
   const int g = a.compGauss[c];
   const float *mu = a.mean + (size_t)g * D, *iv = a.ivar + (size_t)g * D;
   double q = 0.0;
   for (int i = 0; i < D; i++) q += (double)mu[i] * mu[i] * iv[i];
   const double L2E = 1.4426950408889634;
   const float ci = (float)(-0.25 * q * L2E);          // start and closing constant as in model.hip
   const float ce = (float)(((c1 - c0 == 1 ? 0.0 : (double)a.compLogWt[c]) - 0.5 * (double)a.gconst[g] - 0.25 * q) * L2E);
   for (int j = 0; j < 16; j++) { ciRow[j] = ci; endRow[j] = ce; }
   for (int st = 0; st < NS; st++)
      for (int kq = 0; kq < 4; kq++) {
         const int dim = 2 * st + (kq >> 1);
         float v = 0.0f;
         if (dim < D) v = (kq & 1) ? (float)((double)mu[dim] * iv[dim] * L2E) : (float)(-0.5 * (double)iv[dim] * L2E);
         T[(size_t)st * 64 + kq * 16 + col] = v;
      }
}

// the update's counters behind the transition matrices, so that one copy brings both to the host
__global__ void k_upd_export(const int *stats, int *dst, const int *blkStats, int nBlk)
{
   int c0 = 0, c1 = 0, c2 = 0;                               // the fused kernel's per-block counters: stats[0], [1], [5]
   for (int i = threadIdx.x; i < nBlk; i += 64) { c0 += blkStats[3 * i]; c1 += blkStats[3 * i + 1]; c2 += blkStats[3 * i + 2]; }
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
   if (threadIdx.x < 16) dst[threadIdx.x] = stats[threadIdx.x] + (threadIdx.x == 0 ? c0 : threadIdx.x == 1 ? c1 : threadIdx.x == 5 ? c2 : 0);
}

int htkamd_model_refresh_mfma_device(htkamd_model *m, void *stream)
{
   hipStream_t s = (hipStream_t)stream;
   m->mfmaStale = 0;
   if (!m->d_mfmaTab) return HTKAMD_OK;
   MfmaTabArgs t;
   t.D = m->D; t.NS = m->mfmaNS; t.S = m->S; t.stateCompOff = m->d_stateCompOff; t.stateTileOff = m->d_stateTileOff; t.compGauss = m->d_compGauss;
   t.mean = m->d_mean; t.ivar = m->d_ivar; t.gconst = m->d_gconst; t.compLogWt = m->d_compLogWt; t.tab = m->d_mfmaTab;
   const int n = m->nTiles * 16;
   hipLaunchKernelGGL(k_upd_mfma, dim3((n + 255) / 256), dim3(256), 0, s, t, m->nTiles);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// The update in two halves, so that a host loop can queue the next pass behind the update's kernels BEFORE it waits for the few bytes the update
// sends back (transition matrices for the minimum durations, counters): begin = every launch + the copy, recorded in an event; end = wait for
// that event (not for the stream) and fold the copy into the host tables.  htkamd_model_update_device = begin + end.
extern "C" int htkamd_model_update_device_begin(htkamd_model *m, htkamd_accs *accs, const htkamd_update_config *cfg, void *stream)
{
   if (!m || !accs || !cfg) { htkamd_set_error("model_update_device: NULL argument"); return HTKAMD_EINVAL; }
   if (accs->m != m) { htkamd_set_error("model_update_device: accumulators belong to a different model"); return HTKAMD_EINVAL; }
   if (cfg->uFlags & HTKAMD_UPMAP) { htkamd_set_error("model_update_device: MAP re-estimation (HTKAMD_UPMAP) is done by htkamd_model_update"); return HTKAMD_EMODEL; }
   hipStream_t s = (hipStream_t)stream;
   int rc;
   if ((rc = htkamd_model_device_tables(m))) return rc;
   const size_t nFlag = (size_t)m->nT + 2 * (size_t)m->S + 3 * (size_t)m->G;
   const bool tied = m->h_meanLeader != nullptr;
   const size_t nTiedInts = tied ? 2 * (size_t)m->G + (size_t)m->H : 0;           // firstMu[G] firstVa[G] scanPos[H]
   const size_t nLog = tied ? 0 : (((size_t)m->G * m->D + 3) & ~(size_t)3);            // k_upd_gauss_elem's log variances (16-byte aligned block at the end)
   const size_t headBytes = (((nFlag + 63) & ~(size_t)63) + 16 * sizeof(int) + sizeof(float) * (size_t)m->D + sizeof(int) * nTiedInts + 255) & ~(size_t)255;
   const size_t need = headBytes + sizeof(float) * nLog;
   if (need > m->updScratchCap) {
      if (m->d_updScratch) (void)hipFree(m->d_updScratch);
      m->d_updScratch = nullptr; m->updScratchCap = 0;
      HIPCHECK(hipMalloc(&m->d_updScratch, need));
      m->updScratchCap = need;
   }
   unsigned char *fl = (unsigned char *)m->d_updScratch;
   HIPCHECK(hipMemsetAsync(fl, 0, ((nFlag + 63) & ~(size_t)63) + 16 * sizeof(int), s));
   UpdArgs a;
   memset(&a, 0, sizeof(a));
   a.D = m->D; a.S = m->S; a.C = m->C; a.G = m->G; a.nT = m->nT; a.H = m->H; a.PS = m->PS; a.maxM = m->maxM;
   a.stateCompOff = m->d_stateCompOff; a.compGauss = m->d_compGauss; a.transN = m->d_transN; a.transOff = m->d_transOff;
   a.trOccOff = m->d_trOccOff; a.hmmTrans = m->d_hmmTrans; a.hmmStateOff = m->d_hmmStateOff; a.hmmState = m->d_hmmState;
   a.mean = m->d_mean; a.var = m->d_var; a.gconst = m->d_gconst; a.compWeight = m->d_compWeight; a.transP = m->d_transP;
   a.ivar = m->d_ivar; a.gparam = m->d_gparam; a.compLogWt = m->d_compLogWt; a.rawLogWt = m->d_rawLogWt;
   a.acc = accs->d_vec; a.lay = accs->lay;
   a.dimStream = m->NSt > 1 ? m->d_dimStream : nullptr; a.gaussStream = m->NSt > 1 ? m->d_gaussStream : nullptr;
   a.minEgs = cfg->minEgs; a.uFlags = cfg->uFlags; a.singleProcess = m->tiedMix ? 0 : cfg->singleProcess; a.rowNormalise = cfg->rowNormalise;
   a.minVar = cfg->minVar; a.mixWeightFloor = cfg->mixWeightFloor; a.logTpi = log(HTK_TPI);
   a.qualT = fl; a.qualS = a.qualT + m->nT; a.anyS = a.qualS + m->S; a.qualG = a.anyS + m->S; a.anyG = a.qualG + m->G; a.flooredG = a.anyG + m->G;
   a.stats = (int *)(fl + ((nFlag + 63) & ~(size_t)63));
   a.logVar = tied ? nullptr : (float *)(fl + headBytes);
   a.blkStats = (int *)a.logVar;                             // (3 ints per 64 Gaussians in the place of D floats per Gaussian)
   float *dFloor = (float *)(a.stats + 16);
   a.hasVarFloor = cfg->varFloor != nullptr; a.varFloor = dFloor;
   if (cfg->varFloor) HIPCHECK(hipMemcpyAsync(dFloor, cfg->varFloor, sizeof(float) * (size_t)m->D, hipMemcpyHostToDevice, s));
   int *scanPosHost = nullptr;
   if (tied) {
      const int G = m->G;
      a.tied = 1;
      a.meanLeader = m->d_shareTab; a.varLeader = a.meanLeader + G; a.varGroupSize = a.varLeader + G;
      a.muMemOff = a.varGroupSize + G; a.vaMemOff = a.muMemOff + G + 1; a.muMem = a.vaMemOff + G + 1; a.vaMem = a.muMem + m->shareMuMem;
      a.firstMu = (int *)(dFloor + m->D); a.firstVa = a.firstMu + G;
      HIPCHECK(hipMemsetAsync(a.firstMu, 0x7f, sizeof(int) * 2 * (size_t)G, s));         // 0x7f7f7f7f: "no model yet" below
      if (m->h_scanOrder) {
         scanPosHost = (int *)malloc(sizeof(int) * (size_t)m->H);
         for (int k = 0; k < m->H; k++) scanPosHost[m->h_scanOrder[k]] = k;
         HIPCHECK(hipMemcpyAsync(a.firstVa + G, scanPosHost, sizeof(int) * (size_t)m->H, hipMemcpyHostToDevice, s));
         a.scanPos = a.firstVa + G;
      }
   }
   const int B = 128;
   hipLaunchKernelGGL(k_upd_mark, dim3((m->H + B - 1) / B), dim3(B), 0, s, a);
   {  // weights and the marks of the Gaussians, one thread per component; the transition matrices in the launch's last blocks
      const int gw = m->maxM <= 4 ? 4 : m->maxM <= 16 ? 16 : m->maxM <= 64 ? 64 : 0;
      const unsigned nb = gw ? (unsigned)(((size_t)m->S * gw + 255) / 256) : 0, nbT = (unsigned)((m->nT + 255) / 256);
      if (gw == 4) hipLaunchKernelGGL(k_upd_state_w<4>, dim3(nb + nbT), dim3(256), 0, s, a, (int)nb);
      else if (gw == 16) hipLaunchKernelGGL(k_upd_state_w<16>, dim3(nb + nbT), dim3(256), 0, s, a, (int)nb);
      else if (gw == 64) hipLaunchKernelGGL(k_upd_state_w<64>, dim3(nb + nbT), dim3(256), 0, s, a, (int)nb);
      else {
         hipLaunchKernelGGL(k_upd_trans, dim3((m->nT + B - 1) / B), dim3(B), 0, s, a);
         hipLaunchKernelGGL(k_upd_state, dim3((m->S + B - 1) / B), dim3(B), 0, s, a);
      }
   }
   // a tied-mixture set's pool is re-estimated once per set, whatever the models' example counts and the components' weights
   // (MLUpdateModels HERest.c:1272-1279: UpdateTMVars / UpdateTMMeans / FixAllGConsts)
   if (m->tiedMix) HIPCHECK(hipMemsetAsync(a.qualG, 1, (size_t)m->G, s));
   bool fusedG = false;
   {
      const size_t nEl = (size_t)m->G * m->D;
      const size_t ldsFused = sizeof(float) * (size_t)UPD_GPB * (m->D + m->PS) + sizeof(int) * UPD_GPB;
      if (nEl >= ((size_t)1 << 31)) { htkamd_set_error("model_update_device: %zu mean / variance elements (the element kernel indexes with 32 bits)", nEl); return HTKAMD_EMODEL; }
      if (tied) {
         hipLaunchKernelGGL(k_upd_first, dim3((m->H + B - 1) / B), dim3(B), 0, s, a);
         hipLaunchKernelGGL(k_upd_gauss_elem_tied, dim3((unsigned)((nEl + 255) / 256)), dim3(256), 0, s, a);
      } else if (!m->tiedMix && ldsFused <= 60 * 1024 && (m->PS & 3) == 0 && m->D <= 64 && !getenv("HTKAMD_UPD_UNFUSED")) {
         // elements and gConst in one kernel
         const dim3 gr((unsigned)((m->G + UPD_GPB - 1) / UPD_GPB));
         switch ((UPD_GPB * m->D + 1023) / 1024) {
         case 1: hipLaunchKernelGGL(k_upd_gauss_fused<1>, gr, dim3(256), ldsFused, s, a); break;
         case 2: hipLaunchKernelGGL(k_upd_gauss_fused<2>, gr, dim3(256), ldsFused, s, a); break;
         case 3: hipLaunchKernelGGL(k_upd_gauss_fused<3>, gr, dim3(256), ldsFused, s, a); break;
         default: hipLaunchKernelGGL(k_upd_gauss_fused<4>, gr, dim3(256), ldsFused, s, a); break;
         }
         fusedG = true;
      } else
         hipLaunchKernelGGL(k_upd_gauss_elem, dim3((unsigned)((nEl + 1023) / 1024)), dim3(256), 0, s, a);
   }
   if (!fusedG) {
      UpdArgs ag = a;
      if (sizeof(float) * (size_t)B * m->D > 48 * 1024) ag.logVar = nullptr;          // rows too long for the LDS staging: the kernel takes the logs itself
      hipLaunchKernelGGL(k_upd_gconst, dim3((m->G + B - 1) / B), dim3(B), ag.logVar ? sizeof(float) * (size_t)B * m->D : 0, s, ag);
   }
   HIPCHECK(hipGetLastError());
   // the fragment tables of the matrix-core paths that have been scoring with this model now (bf16 x 3, fp16 x 2), the others when
   // they are next asked for (htkamd_launch_score_mfma / _bf16 / _f16)
   m->mfmaStale = 1; m->bf16Stale = 1; m->f16Stale = 1;
   if ((m->fastUse & HTKAMD_SCORE_BF16) && (rc = htkamd_model_refresh_bf16_device(m, s))) return rc;
   if ((m->fastUse & HTKAMD_SCORE_F16) && (rc = htkamd_model_refresh_f16_device(m, s))) return rc;
   m->hostStale = 1;
   // the transition matrices are small and the host needs them (minimum durations for CreateInsts, tee flags for the decoder)
   const size_t nTp = (size_t)m->h_transOff[m->nT];
   if (!m->h_updPin) HIPCHECK(hipHostMalloc(&m->h_updPin, sizeof(float) * (nTp + 16), hipHostMallocDefault));
   if (!m->evUpd) { hipEvent_t e; HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m->evUpd = (void *)e; }
   hipLaunchKernelGGL(k_upd_export, dim3(1), dim3(64), 0, s, a.stats, (int *)(m->d_transP + nTp), a.blkStats, fusedG ? (int)((m->G + UPD_GPB - 1) / UPD_GPB) : 0);
   HIPCHECK(hipMemcpyAsync(m->h_updPin, m->d_transP, sizeof(float) * (nTp + 16), hipMemcpyDeviceToHost, s));
   HIPCHECK(hipEventRecord((hipEvent_t)m->evUpd, s));
   free(scanPosHost);
   m->updPending = 1;
   return HTKAMD_OK;
}

extern "C" int htkamd_model_update_device_end(htkamd_model *m, htkamd_update_stats *stats)
{
   if (!m || !m->updPending) { htkamd_set_error("model_update_device_end: no update in flight"); return HTKAMD_EINVAL; }
   m->updPending = 0;
   HIPCHECK(hipEventSynchronize((hipEvent_t)m->evUpd));
   const size_t nTp = (size_t)m->h_transOff[m->nT];
   int hst[16];
   memcpy(m->h_transP, m->h_updPin, sizeof(float) * nTp);
   memcpy(hst, (const float *)m->h_updPin + nTp, sizeof(hst));
   for (int t = 0; t < m->nT; t++) {
      const int md = htkamd_host_min_dur(m->h_transN[t], m->h_transP + m->h_transOff[t]);
      if (md != m->h_minDur[t]) { m->h_minDur[t] = md; m->topoVersion++; }
      const unsigned char lr = (unsigned char)htkamd_host_trans_is_lr(m->h_transN[t], m->h_transP + m->h_transOff[t]);
      if (lr != m->h_transLR[t]) { m->h_transLR[t] = lr; m->topoVersion++; }
   }
   if (stats) {
      stats->nFloorVar = hst[0]; stats->nFloorVarMix = hst[1]; stats->nSkippedHmm = hst[2];
      stats->nNoTransOut = hst[3]; stats->nNoMixUse = hst[4]; stats->nNoVarUse = hst[5]; stats->nWeightAboveOne = hst[6];
   }
   if (hst[6] > 0) { htkamd_set_error("model_update_device: %d mixture weights above 1.001 (HERest: HError 2393)", hst[6]); return HTKAMD_EMODEL; }
   return HTKAMD_OK;
}

extern "C" int htkamd_model_update_device(htkamd_model *m, htkamd_accs *accs, const htkamd_update_config *cfg, htkamd_update_stats *stats, void *stream)
{
   const int rc = htkamd_model_update_device_begin(m, accs, cfg, stream);
   return rc ? rc : htkamd_model_update_device_end(m, stats);
}
