// gmm_exact.hip -- K1: diagonal-covariance GMM state log-likelihoods, bit-exact to the reference.
//
// Replaces IDOutP (HModel.c:5420-5431) + the mixture loop of ShStrP (HFB.c:949-960) /
// cSOutP (HRec.c:471-491):
//     sum = gConst; for i: xmm = x[i]-mean[i]; sum += xmm*xmm*ivar[i];   (float, in order, no FMA)
//     mixp = -0.5*sum;  x = LAdd(x, wt+mixp)   (float add; LAdd in double, HMath.c:1576; stored to float)
//
// MI355X mapping.  Lanes are FRAMES, the Gaussian is wave-uniform.  One wave owns a tile of 128 frames
// of one utterance and keeps those feature vectors in VGPRs as float2 {frame lane, frame lane+64}, so
// the inner loop is four PACKED FP32 instructions per dimension and Gaussian (v_pk_add, v_pk_mul,
// v_pk_mul, v_pk_add -- gfx950 reaches its FP32 vector rate only with packed instructions; measured
// 75.8 vs 38.5 Tlane-op/s, tools/ubench/valu_rate.hip) with every rounding the reference performs.
// The state/component/Gaussian indices are wave-uniform, so the parameter row ((mean,ivar) pairs,
// 8-byte aligned, then gConst) is fetched with scalar loads through the constant address space and
// feeds the packed instructions as an SGPR pair: no LDS, no VGPRs and no vector-memory traffic for
// parameters.  The mixture log-sum uses the LDS-resident LAdd table (ladd.h).  Tasks (frame tile x
// chunk of chain states, clipped to the utterance's beam) are pulled from a global counter by
// persistent waves, so ragged utterances balance across the 256 CUs.  Output is state-major
// (out[slot*ldo + t]): 64 lanes store 256 contiguous bytes.
//
// Algorithmic work: M*(4*D+8) flop per (frame,state); VALU (packed FP32) bound.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"
#include "ladd.h"

typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float cfloat;    // constant address space -> s_load
typedef const __attribute__((address_space(4))) int cint;

// SOUTP = true: the mixture sum as SOutP forms it (HModel.c:5538-5552) -- bx and px are LogDouble there: the weighted component
// likelihood wt + px is a DOUBLE sum and the running log-sum stays double, rounded to float once at the return -- instead of
// ShStrP's / cSOutP's float after every step.  The two differ in the last bit of ~7 % of scores; tools that call
// OutP / POutP / SOutP directly (HRest, HInit, HVite's direct scoring) see the SOUTP form.
template <int D, bool SOUTP>
__global__ __launch_bounds__(256) void k_score_exact(ScoreArgs a)
{
   __shared__ double tab[LADD_TAB_DOUBLES];
   ladd_table_to_lds(tab, a.laddTab);
   __syncthreads();
   const int lane = threadIdx.x & 63;
   const double mle = a.minLogExp;
   cint *slotState = (cint *)a.slotState;
   cint *stateCompOff = (cint *)a.stateCompOff;
   cint *compGauss = (cint *)a.compGauss;
   cfloat *compLogWt = (cfloat *)a.compLogWt;

   for (;;) {
      int task = 0;
      if (lane == 0) task = atomicAdd(a.taskCounter, 1);
      task = __builtin_amdgcn_readfirstlane(task);
      if (task >= a.nTasks) break;
      const ScoreTask tk = a.tasks[task];

      v2f x[D];
      {
         int t0 = lane, t1 = lane + 64;
         if (t0 > tk.nFrames - 1) t0 = tk.nFrames - 1;
         if (t1 > tk.nFrames - 1) t1 = tk.nFrames - 1;
         const float *r0 = a.X + (size_t)(tk.frame0 + t0) * D;
         const float *r1 = a.X + (size_t)(tk.frame0 + t1) * D;
#pragma unroll
         for (int i = 0; i < D; i++) { x[i].x = r0[i]; x[i].y = r1[i]; }
      }

      for (int k = 0; k < tk.nSlots; k++) {
         const int s = slotState[tk.slot0 + k];
         const int c0 = stateCompOff[s], c1 = stateCompOff[s + 1];
         float acc0, acc1;
         if (c1 - c0 == 1) {                    // single Gaussian: no weight, no LAdd (HFB.c:917-928)
            cfloat *P = (cfloat *)(a.gparam + (size_t)compGauss[c0] * a.PS);
            v2f sum = {P[2 * D], P[2 * D]};
#pragma unroll
            for (int i = 0; i < D; i++) {
               const float mu = P[2 * i], iv = P[2 * i + 1];
               v2f xm = x[i] - mu;
               xm = xm * xm;
               xm = xm * iv;
               sum = sum + xm;
            }
            acc0 = -0.5f * sum.x; acc1 = -0.5f * sum.y;
         } else {
            acc0 = (float)LZERO; acc1 = (float)LZERO;
            double dacc0 = LZERO, dacc1 = LZERO;
            for (int c = c0; c < c1; c++) {
               const float wt = compLogWt[c];
               if (wt > (float)LMINMIX) {       // wave-uniform branch
                  cfloat *P = (cfloat *)(a.gparam + (size_t)compGauss[c] * a.PS);
#ifdef EX_TRUTH                                         /* diagnostic build: float64 arithmetic on the same fp32 parameters -- what a scorer without rounding error
                                                           would hand the recursions (tools/r06_parvar.sh: the floor of the tolerance class against the reference's floats) */
                  {
                     double s0 = P[2 * D], s1 = P[2 * D];
#pragma unroll
                     for (int i = 0; i < D; i++) {
                        const double mu = P[2 * i], iv = P[2 * i + 1];
                        const double d0 = (double)x[i].x - mu, d1 = (double)x[i].y - mu;
                        s0 += d0 * d0 * iv; s1 += d1 * d1 * iv;
                     }
                     dacc0 = ladd_tab(dacc0, (double)wt - 0.5 * s0, mle, tab);
                     dacc1 = ladd_tab(dacc1, (double)wt - 0.5 * s1, mle, tab);
                     acc0 = (float)dacc0; acc1 = (float)dacc1;
                     continue;
                  }
#endif
                  v2f sum = {P[2 * D], P[2 * D]};
#pragma unroll
                  for (int i = 0; i < D; i++) {
                     const float mu = P[2 * i], iv = P[2 * i + 1];
                     v2f xm = x[i] - mu;
                     xm = xm * xm;
                     xm = xm * iv;
                     sum = sum + xm;
                  }
                  if constexpr (SOUTP) {
                     const v2f px = -0.5f * sum;
                     dacc0 = ladd_tab(dacc0, (double)wt + (double)px.x, mle, tab);
                     dacc1 = ladd_tab(dacc1, (double)wt + (double)px.y, mle, tab);
                  } else {
                     const v2f y = wt + (-0.5f * sum);
#ifdef EX_ABL_NOLADD
                     acc0 = fmaxf(acc0, y.x); acc1 = fmaxf(acc1, y.y);
#elif defined(EX_LADD_TWICE)
                     acc0 = ladd_tab_f(acc0, y.x, mle, tab);
                     acc1 = ladd_tab_f(acc1, y.y, mle, tab);
#else
                     ladd_tab_f2(acc0, y.x, acc1, y.y, mle, tab);
#endif
                  }
               }
            }
            if constexpr (SOUTP) { acc0 = (float)dacc0; acc1 = (float)dacc1; }
         }
         float *o = a.out + tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo;
         if (lane < tk.nFrames) o[lane] = acc0;
         if (lane + 64 < tk.nFrames) o[lane + 64] = acc1;
      }
   }
}

// Any vector size: features are re-read from global memory per dimension (L1-resident rows).
// DIAGC = true: DOutP's form (HModel.c:5347-5358), xmm*xmm/var with the float division, for sets that were not put through
// ConvDiagC (HRest, HInit and every other direct caller of OutP score DIAGC sets as they were loaded).
template <bool SOUTP, bool DIAGC>
__global__ __launch_bounds__(256) void k_score_exact_anyD(ScoreArgs a)
{
   __shared__ double tab[LADD_TAB_DOUBLES];
   ladd_table_to_lds(tab, a.laddTab);
   __syncthreads();
   const int lane = threadIdx.x & 63;
   const int D = a.D;
   for (;;) {
      int task = 0;
      if (lane == 0) task = atomicAdd(a.taskCounter, 1);
      task = __builtin_amdgcn_readfirstlane(task);
      if (task >= a.nTasks) break;
      const ScoreTask tk = a.tasks[task];
      for (int f = 0; f < 2; f++) {
         int t = lane + 64 * f;
         const bool live = t < tk.nFrames;
         if (!live) t = tk.nFrames - 1;
         const float *row = a.X + (size_t)(tk.frame0 + t) * D;
         for (int k = 0; k < tk.nSlots; k++) {
            const int s0 = a.slotState[tk.slot0 + k];
            float total = 0.0f, acc = (float)LZERO;
            for (int ks = 0; ks < (a.NSt > 1 ? a.NSt : 1); ks++) {
            const int s = s0 + ks;
            const int c0 = a.stateCompOff[s], c1 = a.stateCompOff[s + 1];
            acc = (float)LZERO;
            double dacc = LZERO;
            for (int c = c0; c < c1; c++) {
               const float wt = a.compLogWt[c];
               if (c1 - c0 > 1 && !(wt > (float)LMINMIX)) continue;
               const float *P = a.gparam + (size_t)a.compGauss[c] * a.PS;
               float sum = P[2 * D];
               if constexpr (DIAGC) {
                  const float *V = a.var + (size_t)a.compGauss[c] * D;
                  for (int i = 0; i < D; i++) {
                     float xmm = row[i] - P[2 * i];
                     sum += xmm * xmm / V[i];
                  }
               } else
               for (int i = 0; i < D; i++) {
                  float xmm = row[i] - P[2 * i];
                  sum += xmm * xmm * P[2 * i + 1];
               }
               float mixp = -0.5f * sum;
               if (c1 - c0 == 1) acc = mixp;
               else if constexpr (SOUTP) dacc = ladd_tab(dacc, (double)wt + (double)mixp, a.minLogExp, tab);
               else {
                  float y = wt + mixp;
                  acc = ladd_tab_f(acc, y, a.minLogExp, tab);
               }
            }
            if (SOUTP && c1 - c0 > 1) acc = (float)dacc;
            if (a.NSt > 1) { const float wx = a.streamWt[s] * acc; total = total + wx; }     // cPOutP / POutP: float product, float sum
            }
            if (a.NSt > 1) acc = total;
            if (live) a.out[tk.outBase + (size_t)(tk.outSlot0 + k) * tk.ldo + t] = acc;
         }
      }
   }
}

int htkamd_launch_score_exact(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop, bool soutp, bool diagc)
{
   if (a.nTasks <= 0) return HTKAMD_OK;
   HIPCHECK(hipMemsetAsync(a.taskCounter, 0, sizeof(int), stream));
   int blocks = (a.nTasks + 3) / 4;
   if (blocks > 256 * 5) blocks = 256 * 5;      // persistent: up to 5 four-wave blocks per CU (VGPR-limited)
   dim3 grid(blocks), block(256);
   if (a.NSt > 1 && !a.streamWt) { htkamd_set_error("score_exact: several streams without stream weights"); return HTKAMD_EINVAL; }
   if (a.NSt > 1 && !soutp && !diagc) {          // stream-weighted state probabilities (HVite on a multi-stream set): the general kernel
      hipExtLaunchKernelGGL((k_score_exact_anyD<false, false>), grid, block, 0, stream, evStart, evStop, 0, a);
      HIPCHECK(hipGetLastError());
      return HTKAMD_OK;
   }
   if (soutp || diagc) {                         // the HRest / HInit / direct-OutP forms: not a hot path, one kernel for every size
      if (diagc && !a.var) { htkamd_set_error("score_exact: DIAGC form without the variance table"); return HTKAMD_EINVAL; }
      if (soutp && diagc) hipExtLaunchKernelGGL((k_score_exact_anyD<true, true>), grid, block, 0, stream, evStart, evStop, 0, a);
      else if (soutp) hipExtLaunchKernelGGL((k_score_exact_anyD<true, false>), grid, block, 0, stream, evStart, evStop, 0, a);
      else hipExtLaunchKernelGGL((k_score_exact_anyD<false, true>), grid, block, 0, stream, evStart, evStop, 0, a);
      HIPCHECK(hipGetLastError());
      return HTKAMD_OK;
   }
   switch (m->D) {
   case 39: hipExtLaunchKernelGGL((k_score_exact<39, false>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 26: hipExtLaunchKernelGGL((k_score_exact<26, false>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   case 13: hipExtLaunchKernelGGL((k_score_exact<13, false>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   default: hipExtLaunchKernelGGL((k_score_exact_anyD<false, false>), grid, block, 0, stream, evStart, evStop, 0, a); break;
   }
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ C ABI
static int outp_block(htkamd_model *m, const float *dX, int T, const int *dStates, int ns, float *dOut, int ldo, int mode, void *stream);

extern "C" int htkamd_outp_block(htkamd_model *m, const float *dX, int T, const int *dStates, int ns,
                                 float *dOut, int ldo, void *stream)
{
   return outp_block(m, dX, T, dStates, ns, dOut, ldo, HTKAMD_SCORE_EXACT, stream);
}

extern "C" int htkamd_outp_block_mode(htkamd_model *m, const float *dX, int T, const int *dStates, int ns,
                                      float *dOut, int ldo, int scoreMode, void *stream)
{
   if (scoreMode != HTKAMD_SCORE_EXACT && scoreMode != HTKAMD_SCORE_MFMA && scoreMode != HTKAMD_SCORE_BF16 && scoreMode != HTKAMD_SCORE_F16 && (scoreMode & ~(HTKAMD_SCORE_SOUTP | HTKAMD_SCORE_DIAGC))) { htkamd_set_error("outp_block: unknown score mode %d", scoreMode); return HTKAMD_EINVAL; }
   return outp_block(m, dX, T, dStates, ns, dOut, ldo, scoreMode, stream);
}

// Task tables of outp_block calls live in a small ring of persistent (device, pinned host) buffer pairs owned by the model: no
// allocation, no free and no synchronisation per call -- a caller that scores a buffer of frames per call (HVite's POutP through the
// HTKLib shim, HDecode's block scorer) pays a launch, not an allocator round trip.  A slot is reused only after the event recorded
// behind its kernel has completed, so up to OB_SLOTS calls may be in flight (on any streams).
#define OB_SLOTS 4
struct ObSlot { char *d, *h; size_t cap; hipEvent_t ev; bool hasEv, busy; };
struct ObRing { ObSlot s[OB_SLOTS]; int next; };

void htkamd_outp_ring_free(void *ring)
{
   ObRing *r = (ObRing *)ring;
   if (!r) return;
   for (int i = 0; i < OB_SLOTS; i++) {
      if (r->s[i].busy) (void)hipEventSynchronize(r->s[i].ev);
      if (r->s[i].d) (void)hipFree(r->s[i].d);
      if (r->s[i].h) (void)hipHostFree(r->s[i].h);
      if (r->s[i].hasEv) (void)hipEventDestroy(r->s[i].ev);
   }
   free(r);
}

static int outp_block(htkamd_model *m, const float *dX, int T, const int *dStates, int ns, float *dOut, int ldo, int mode, void *stream)
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
   if (!m->obRing) { m->obRing = calloc(1, sizeof(ObRing)); if (!m->obRing) { htkamd_set_error("outp_block: out of memory"); return HTKAMD_ENOMEM; } }
   ObRing *ring = (ObRing *)m->obRing;
   ObSlot &sl = ring->s[ring->next];
   ring->next = (ring->next + 1) % OB_SLOTS;
   if (!sl.hasEv) { HIPCHECK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming)); sl.hasEv = true; }
   if (sl.busy) { HIPCHECK(hipEventSynchronize(sl.ev)); sl.busy = false; }        // the call that used this slot four calls ago
   const size_t need = sizeof(ScoreTask) * (size_t)nTasks + sizeof(int);
   if (need > sl.cap) {
      if (sl.d) (void)hipFree(sl.d);
      if (sl.h) (void)hipHostFree(sl.h);
      sl.d = sl.h = nullptr; sl.cap = 0;
      const size_t want = need + need / 2 + 4096;
      HIPCHECK(hipMalloc((void **)&sl.d, want));
      HIPCHECK(hipHostMalloc((void **)&sl.h, want, hipHostMallocDefault));
      sl.cap = want;
   }
   ScoreTask *h = (ScoreTask *)sl.h;
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
   HIPCHECK(hipMemcpyAsync(sl.d, sl.h, sizeof(ScoreTask) * (size_t)nTasks, hipMemcpyHostToDevice, s));
   ScoreArgs a;
   a.tasks = (const ScoreTask *)sl.d; a.nTasks = nTasks; a.X = dX; a.slotState = dStates; a.out = dOut;
   a.stateCompOff = m->d_stateCompOff; a.compGauss = m->d_compGauss; a.compLogWt = m->d_compLogWt;
   a.gparam = m->d_gparam; a.PS = m->PS; a.D = m->D; a.minLogExp = m->minLogExp;
   a.laddTab = m->d_laddTab; a.taskCounter = (int *)(sl.d + sizeof(ScoreTask) * (size_t)nTasks);
   a.mfmaTab = m->d_mfmaTab; a.stateTileOff = m->d_stateTileOff; a.bf16Tab = m->d_bf16Tab;
   if (mode & HTKAMD_SCORE_DIAGC) { int rcv = htkamd_model_device_tables(m); if (rcv) return rcv; }
   a.var = m->d_var;
   const int rc = htkamd_launch_score(mode, m, a, s);
   HIPCHECK(hipEventRecord(sl.ev, s));
   sl.busy = true;
   return rc;
}
