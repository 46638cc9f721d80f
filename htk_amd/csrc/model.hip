// model.hip -- packed HMM set on the device, accumulator vector, error plumbing.
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "internal.h"
#include "hipcheck.h"

static thread_local char g_err[512] = "";

extern "C" void htkamd_set_error(const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof(g_err), fmt, ap);
   va_end(ap);
}

extern "C" const char *htkamd_last_error(void) { return g_err; }
extern "C" int htkamd_version(void) { return 100; }

extern "C" int htkamd_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

extern "C" int htkamd_set_device(int ordinal)
{
   HIPCHECK(hipSetDevice(ordinal));
   return HTKAMD_OK;
}

extern "C" int htkamd_dev_malloc(void **dptr, size_t bytes)
{
   if (!dptr) { htkamd_set_error("dev_malloc: NULL"); return HTKAMD_EINVAL; }
   hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
   if (e != hipSuccess) { htkamd_set_error("dev_malloc(%zu): %s", bytes, hipGetErrorString(e)); return (e == hipErrorOutOfMemory) ? HTKAMD_ENOMEM : HTKAMD_EHIP; }
   return HTKAMD_OK;
}

extern "C" int htkamd_dev_free(void *dptr)
{
   if (dptr) HIPCHECK(hipFree(dptr));
   return HTKAMD_OK;
}

extern "C" int htkamd_memcpy_h2d(void *dDst, const void *hSrc, size_t bytes, void *stream)
{
   if (bytes == 0) return HTKAMD_OK;
   HIPCHECK(hipMemcpyAsync(dDst, hSrc, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
   HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
   return HTKAMD_OK;
}

// host memory the device can read directly (page-locked): a copy from it is a plain DMA, and htkamd_memcpy_h2d_async need not wait for it --
// the copy is ordered before whatever follows on the same stream; the buffer may be refilled once that work has been waited for
extern "C" int htkamd_host_malloc(void **hptr, size_t bytes)
{
   if (!hptr) { htkamd_set_error("host_malloc: NULL"); return HTKAMD_EINVAL; }
   hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
   if (e != hipSuccess) { htkamd_set_error("host_malloc(%zu): %s", bytes, hipGetErrorString(e)); return (e == hipErrorOutOfMemory) ? HTKAMD_ENOMEM : HTKAMD_EHIP; }
   return HTKAMD_OK;
}

extern "C" int htkamd_host_free(void *hptr)
{
   if (hptr) HIPCHECK(hipHostFree(hptr));
   return HTKAMD_OK;
}

extern "C" int htkamd_memcpy_h2d_async(void *dDst, const void *hSrc, size_t bytes, void *stream)
{
   if (bytes == 0) return HTKAMD_OK;
   HIPCHECK(hipMemcpyAsync(dDst, hSrc, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
   return HTKAMD_OK;
}

extern "C" int htkamd_memcpy_d2h(void *hDst, const void *dSrc, size_t bytes, void *stream)
{
   if (bytes == 0) return HTKAMD_OK;
   HIPCHECK(hipMemcpyAsync(hDst, dSrc, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
   HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
   return HTKAMD_OK;
}

extern "C" int htkamd_stream_sync(void *stream)
{
   HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
   return HTKAMD_OK;
}

extern "C" int htkamd_stream_create(void **stream)
{
   if (!stream) { htkamd_set_error("stream_create: NULL"); return HTKAMD_EINVAL; }
   hipStream_t s;
   HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
   *stream = (void *)s;
   return HTKAMD_OK;
}

extern "C" int htkamd_stream_destroy(void *stream)
{
   if (stream) HIPCHECK(hipStreamDestroy((hipStream_t)stream));
   return HTKAMD_OK;
}

// what is queued on `waiter` from here on starts behind what has been queued on `signaller` so far
extern "C" int htkamd_stream_wait(void *waiter, void *signaller)
{
   hipEvent_t ev;
   HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
   hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
   if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
   (void)hipEventDestroy(ev);                            // (released when the record has completed)
   if (e != hipSuccess) { htkamd_set_error("stream_wait: %s", hipGetErrorString(e)); return HTKAMD_EHIP; }
   return HTKAMD_OK;
}

template <typename T> static T *dupHost(const T *src, size_t n)
{
   T *p = (T *)malloc(sizeof(T) * (n ? n : 1));
   if (src && n) memcpy(p, src, sizeof(T) * n);
   return p;
}

template <typename T> static int toDevice(T **dst, const T *src, size_t n)
{
   if (*dst == nullptr) HIPCHECK(hipMalloc((void **)dst, sizeof(T) * (n ? n : 1)));
   if (n) HIPCHECK(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
   return HTKAMD_OK;
}

// A-operand fragment table of the MFMA scoring kernel (layout: gmm_mfma.hip).  Column j of tile t of state s is
// component stateCompOff[s] + 16*(t - stateTileOff[s]) + j; K index 4*step + kq carries dimension 2*step + (kq>>1),
// as the x^2 coefficient -0.5*ivar for even kq and the x coefficient mean*ivar for odd kq; rows mfmaNS..mfmaNS+3 are
// the accumulator start -0.25*sum mean^2*ivar and rows mfmaNS+4..mfmaNS+7 the closing constant log weight - 0.5*gConst -
// 0.25*sum mean^2*ivar, both in the C-operand layout.  Unused components close at -1e30 (drop out of the sum).
static int mfma_refresh(htkamd_model *m)
{
   const int D = m->D;
   if (D > 48) return HTKAMD_OK;                      // no matrix-core kernel: the exact path serves such sets
   // fp32 path: K steps of 4 = 2 dimensions each; kernels exist for 7, 13 and 20 steps, smaller sizes are zero-padded up to the next
   const int need = (D + 1) / 2;
   const int NS = need <= 7 ? 7 : (need <= 13 ? 13 : 20);
   if (!m->d_stateTileOff) {
      int *off = (int *)malloc(sizeof(int) * ((size_t)m->S + 1));
      off[0] = 0;
      for (int s = 0; s < m->S; s++) off[s + 1] = off[s] + (m->h_stateCompOff[s + 1] - m->h_stateCompOff[s] + 15) / 16;
      m->nTiles = off[m->S]; m->mfmaNS = NS;
      int rc = toDevice(&m->d_stateTileOff, off, (size_t)m->S + 1);
      if (!rc) {
         int *ts = (int *)malloc(sizeof(int) * (size_t)(m->nTiles ? m->nTiles : 1));
         for (int s = 0; s < m->S; s++) for (int t = off[s]; t < off[s + 1]; t++) ts[t] = s;
         rc = toDevice(&m->d_tileState, ts, (size_t)m->nTiles);
         free(ts);
      }
      free(off);
      if (rc) return rc;
      m->bf16NC = (D + 14) / 15;                      // 15 dimensions as (x^2, x) pairs + the chunk's constant per K chunk of 32 (gmm_bf16.hip)
      if (m->bf16NC <= 3) {
         HIPCHECK(hipMalloc(&m->d_bf16Tab, (size_t)m->nTiles * ((size_t)3 * m->bf16NC * 64 * 16 + 64 * 16)));
         m->f16Wide = m->nTiles == m->S && !getenv("HTKAMD_F16_NARROW");      // (the switch: tests of the 16 x 16 form on such sets)
         m->bf16Dense = m->f16Wide && D >= 31 && D <= 39 && !getenv("HTKAMD_BF16_CHUNKED");      // (the switch: the six-k-step layout on such sets, for comparisons)
         HIPCHECK(hipMalloc(&m->d_f16Tab, (size_t)m->nTiles * ((size_t)2 * m->bf16NC * 64 * 16 + 64 * 16)));
         HIPCHECK(hipMalloc(&m->d_f16Ctl, sizeof(float) * 512));
         HIPCHECK(hipMemset(m->d_f16Ctl, 0, sizeof(float) * 512));
      }
   }
   {  // bf16 x 3 and fp16 x 2 paths: their tables are built on the device from the tables just uploaded
      int rcb = htkamd_model_refresh_bf16_device(m, nullptr);
      if (!rcb) rcb = htkamd_model_refresh_f16_device(m, nullptr);
      if (rcb) return rcb;
      HIPCHECK(hipStreamSynchronize(nullptr));
   }
   if (D > 40) return HTKAMD_OK;
   const size_t stride = (size_t)(NS + 8) * 64;
   float *tab = (float *)calloc((size_t)m->nTiles * stride, sizeof(float));
   size_t t = 0;
   for (int s = 0; s < m->S; s++) {
      const int c0 = m->h_stateCompOff[s], c1 = m->h_stateCompOff[s + 1];
      for (int cb = c0; cb < c1; cb += 16, t++) {
         float *T = tab + t * stride;
         for (int col = 0; col < 16; col++) {
            const int c = cb + col;
            const bool live = c < c1 && (c1 - c0 == 1 || m->h_compLogWt[c] > (float)LMINMIX);
            // accumulator start of row `col`: lane (kq = col/4, any column) register col%4, i.e. table row NS + col%4
            float *ciRow = T + (size_t)(NS + (col & 3)) * 64 + (col >> 2) * 16, *endRow = ciRow + 4 * 64;
            if (!live) { for (int j = 0; j < 16; j++) { ciRow[j] = 0.0f; endRow[j] = -1.0e30f; } continue; }
            const int g = m->h_compGauss[c];
            const float *mu = m->h_mean + (size_t)g * D, *iv = m->h_ivar + (size_t)g * D;
            double q = 0.0;
            for (int i = 0; i < D; i++) q += (double)mu[i] * mu[i] * iv[i];
            const double L2E = 1.4426950408889634;           // table in base-2 logarithms (gmm_mfma.hip)
            // the accumulators start at half of -0.5 sum mu^2 ivar; the rest is added after the contraction (gmm_mfma.hip)
            const float ci = (float)(-0.25 * q * L2E);
            const float ce = (float)(((c1 - c0 == 1 ? 0.0 : (double)m->h_compLogWt[c]) - 0.5 * (double)m->h_gconst[g] - 0.25 * q) * L2E);
            for (int j = 0; j < 16; j++) { ciRow[j] = ci; endRow[j] = ce; }
            for (int st = 0; st < NS; st++)
               for (int kq = 0; kq < 4; kq++) {
                  const int dim = 2 * st + (kq >> 1);
                  float v = 0.0f;
                  if (dim < D) v = (kq & 1) ? (float)((double)mu[dim] * iv[dim] * L2E) : (float)(-0.5 * (double)iv[dim] * L2E);
                  T[(size_t)st * 64 + kq * 16 + col] = v;
               }
         }
      }
   }
   int rc = toDevice(&m->d_mfmaTab, tab, (size_t)m->nTiles * stride);
   free(tab);
   m->mfmaStale = 0;
   return rc;
}

// (Re)derive ivar / log weights / min durations / the interleaved scoring table and push them.
static int model_refresh(htkamd_model *m, bool derive = true)
{
   const int D = m->D, PS = m->PS;
   if (derive) {
      htkamd_host_conv_diagc((size_t)m->G * D, m->h_var, m->h_ivar);
      if (m->NSt > 1)                                    // a dimension outside the Gaussian's stream takes no part in its score
         for (int g = 0; g < m->G; g++)
            for (int k = 0; k < D; k++) if (m->h_dimStream[k] != m->h_gaussStream[g]) m->h_ivar[(size_t)g * D + k] = 0.0f;
      for (int c = 0; c < m->C; c++)
         m->h_compLogWt[c] = (m->h_rawLogWt && m->h_rawLogWt[c]) ? m->h_compWeight[c] : htkamd_host_mix_log_weight(m->h_compWeight[c]);
   }
   for (int t = 0; t < m->nT; t++) {
      const int md = htkamd_host_min_dur(m->h_transN[t], m->h_transP + m->h_transOff[t]);
      if (md != m->h_minDur[t]) { m->h_minDur[t] = md; m->topoVersion++; }
      const unsigned char lr = (unsigned char)htkamd_host_trans_is_lr(m->h_transN[t], m->h_transP + m->h_transOff[t]);
      if (lr != m->h_transLR[t]) { m->h_transLR[t] = lr; m->topoVersion++; }
   }
   float *gp = (float *)calloc((size_t)m->G * PS, sizeof(float));
   for (int g = 0; g < m->G; g++) {
      float *p = gp + (size_t)g * PS;
      for (int i = 0; i < D; i++) {
         p[2 * i] = m->h_mean[(size_t)g * D + i];
         p[2 * i + 1] = m->h_ivar[(size_t)g * D + i];
      }
      p[2 * D] = m->h_gconst[g];
   }
   int rc = toDevice(&m->d_gparam, gp, (size_t)m->G * PS);
   free(gp);
   if (rc) return rc;
   if ((rc = toDevice(&m->d_mean, m->h_mean, (size_t)m->G * D))) return rc;
   if ((rc = toDevice(&m->d_ivar, m->h_ivar, (size_t)m->G * D))) return rc;
   if ((rc = toDevice(&m->d_gconst, m->h_gconst, (size_t)m->G))) return rc;
   if ((rc = toDevice(&m->d_compLogWt, m->h_compLogWt, (size_t)m->C))) return rc;
   if (!m->d_transP) HIPCHECK(hipMalloc((void **)&m->d_transP, sizeof(float) * ((size_t)m->h_transOff[m->nT] + 16)));      // + the device update's 16 counters: one copy brings both back (update.hip)
   if ((rc = toDevice(&m->d_transP, m->h_transP, (size_t)m->h_transOff[m->nT]))) return rc;
   if (m->d_var && ((rc = toDevice(&m->d_var, m->h_var, (size_t)m->G * D)) || (rc = toDevice(&m->d_compWeight, m->h_compWeight, (size_t)m->C)))) return rc;
   return mfma_refresh(m);
}

// Linear parameters and the model topology on the device, for the device-side update (uploaded on its first call).
int htkamd_model_device_tables(htkamd_model *m)
{
   if (m->d_var) return HTKAMD_OK;
   int rc;
   if ((rc = toDevice(&m->d_var, m->h_var, (size_t)m->G * m->D)) || (rc = toDevice(&m->d_compWeight, m->h_compWeight, (size_t)m->C)) ||
       (rc = toDevice(&m->d_trOccOff, m->h_trOccOff, (size_t)m->nT + 1)) || (rc = toDevice(&m->d_hmmTrans, m->h_hmmTrans, (size_t)m->H)) ||
       (rc = toDevice(&m->d_hmmStateOff, m->h_hmmStateOff, (size_t)m->H + 1)) ||
       (rc = toDevice(&m->d_hmmState, m->h_hmmState, (size_t)m->h_hmmStateOff[m->H]))) return rc;
   return HTKAMD_OK;
}

// After a device-side update the host copies are refreshed on demand (get_params, the host update, set_params).
int htkamd_model_sync_host(htkamd_model *m)
{
   if (!m->hostStale) return HTKAMD_OK;
   HIPCHECK(hipMemcpy(m->h_mean, m->d_mean, sizeof(float) * (size_t)m->G * m->D, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(m->h_var, m->d_var, sizeof(float) * (size_t)m->G * m->D, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(m->h_ivar, m->d_ivar, sizeof(float) * (size_t)m->G * m->D, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(m->h_gconst, m->d_gconst, sizeof(float) * (size_t)m->G, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(m->h_compWeight, m->d_compWeight, sizeof(float) * (size_t)m->C, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(m->h_compLogWt, m->d_compLogWt, sizeof(float) * (size_t)m->C, hipMemcpyDeviceToHost));
   m->hostStale = 0;
   return HTKAMD_OK;
}

extern "C" int htkamd_model_create(const htkamd_model_desc *d, htkamd_model **out)
{
   if (!d || !out) { htkamd_set_error("model_create: NULL argument"); return HTKAMD_EINVAL; }
   if (htkamd_device_count() <= 0) { htkamd_set_error("model_create: no HIP device"); return HTKAMD_ENODEV; }
   if (d->vecSize <= 0 || d->numStates <= 0 || d->numComp <= 0 || d->numGauss <= 0 || d->numTrans <= 0 || d->numPhys <= 0) {
      htkamd_set_error("model_create: empty model"); return HTKAMD_EINVAL;
   }
   htkamd_model *m = (htkamd_model *)calloc(1, sizeof(htkamd_model));
   const int NSt = d->numStreams > 1 ? d->numStreams : 1;
   if (NSt > 1 && !d->dimStream) { free(m); htkamd_set_error("model_create: numStreams = %d without dimStream", NSt); return HTKAMD_EINVAL; }
   m->NSt = NSt;
   m->D = d->vecSize; m->S = d->numStates * NSt; m->C = d->numComp; m->G = d->numGauss; m->nT = d->numTrans; m->H = d->numPhys;
   m->PS = ((2 * m->D + 1) + 3) & ~3;
   m->minLogExp = htkamd_host_min_log_exp();
   m->h_stateCompOff = dupHost(d->stateCompOff, (size_t)m->S + 1);
   m->h_compGauss = dupHost(d->compGauss, (size_t)m->C);
   m->h_compWeight = dupHost(d->compWeight, (size_t)m->C);
   m->h_compLogWt = dupHost((const float *)nullptr, (size_t)m->C);
   m->h_mean = dupHost(d->mean, (size_t)m->G * m->D);
   m->h_var = dupHost(d->var, (size_t)m->G * m->D);
   m->h_ivar = dupHost((const float *)nullptr, (size_t)m->G * m->D);
   m->h_transN = dupHost(d->transN, (size_t)m->nT);
   m->h_transOff = dupHost(d->transOff, (size_t)m->nT + 1);
   m->h_transP = dupHost(d->transP, (size_t)d->transOff[m->nT]);
   m->h_hmmTrans = dupHost(d->hmmTrans, (size_t)m->H);
   m->h_hmmStateOff = dupHost(d->hmmStateOff, (size_t)m->H + 1);
   m->h_hmmState = dupHost(d->hmmState, (size_t)d->hmmStateOff[m->H]);
   if (NSt > 1) {                                        // every emitting state becomes its NSt elements
      const int n = d->hmmStateOff[m->H];
      free(m->h_hmmState);
      m->h_hmmState = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1) * NSt);
      for (int i = 0; i < n; i++) for (int k = 0; k < NSt; k++) m->h_hmmState[(size_t)i * NSt + k] = d->hmmState[i] * NSt + k;
      for (int h = 0; h <= m->H; h++) m->h_hmmStateOff[h] = d->hmmStateOff[h] * NSt;
      m->h_dimStream = dupHost(d->dimStream, (size_t)m->D);
      m->h_gaussStream = (int *)malloc(sizeof(int) * (size_t)m->G);
      for (int g = 0; g < m->G; g++) m->h_gaussStream[g] = -1;
      for (int k = 0; k < m->D; k++) if (d->dimStream[k] < 0 || d->dimStream[k] >= NSt) { htkamd_set_error("model_create: dimStream[%d] = %d", k, d->dimStream[k]); htkamd_model_destroy(m); return HTKAMD_EINVAL; }
      for (int e = 0; e < m->S; e++)
         for (int c = d->stateCompOff[e]; c < d->stateCompOff[e + 1]; c++) {
            const int g = d->compGauss[c];
            if (m->h_gaussStream[g] >= 0 && m->h_gaussStream[g] != e % NSt) { htkamd_set_error("model_create: Gaussian %d is used by streams %d and %d", g, m->h_gaussStream[g] + 1, e % NSt + 1); htkamd_model_destroy(m); return HTKAMD_EINVAL; }
            m->h_gaussStream[g] = e % NSt;
         }
      for (int g = 0; g < m->G; g++) if (m->h_gaussStream[g] < 0) m->h_gaussStream[g] = 0;
      m->h_streamWt = (float *)malloc(sizeof(float) * (size_t)m->S);
      for (int e = 0; e < m->S; e++) m->h_streamWt[e] = d->streamWeight ? d->streamWeight[e] : 1.0f;
   }
   m->h_minDur = dupHost((const int *)nullptr, (size_t)m->nT);
   for (int t = 0; t < m->nT; t++) m->h_minDur[t] = -1;
   m->h_transLR = (unsigned char *)malloc((size_t)(m->nT ? m->nT : 1));
   for (int t = 0; t < m->nT; t++) m->h_transLR[t] = 2;          // unknown: model_refresh sets it
   m->h_trOccOff = dupHost((const int *)nullptr, (size_t)m->nT + 1);
   m->h_gconst = dupHost(d->gconst, (size_t)m->G);
   if (!d->gconst)                                   // CheckMix: gConst fixed at load (HModel.c:206-208)
      for (int g = 0; g < m->G; g++) htkamd_host_fix_diag_gconst_ms(m->D, m->h_var + (size_t)g * m->D, m->h_dimStream, m->h_gaussStream ? m->h_gaussStream[g] : 0, m->h_gconst + g);
   m->maxN = 0; m->maxM = 1; m->h_trOccOff[0] = 0;
   for (int t = 0; t < m->nT; t++) {
      if (m->h_transN[t] > m->maxN) m->maxN = m->h_transN[t];
      m->h_trOccOff[t + 1] = m->h_trOccOff[t] + m->h_transN[t];
      if (m->h_transOff[t + 1] - m->h_transOff[t] != m->h_transN[t] * m->h_transN[t]) {
         htkamd_set_error("model_create: transOff inconsistent with transN at matrix %d", t);
         htkamd_model_destroy(m); return HTKAMD_EINVAL;
      }
   }
   for (int s = 0; s < m->S; s++) {
      int M = m->h_stateCompOff[s + 1] - m->h_stateCompOff[s];
      if (M < 1) { htkamd_set_error("model_create: state %d has no mixture component", s); htkamd_model_destroy(m); return HTKAMD_EINVAL; }
      if (M > m->maxM) m->maxM = M;
   }
   for (int h = 0; h < m->H; h++) {
      int N = m->h_transN[m->h_hmmTrans[h]];
      if (m->h_hmmStateOff[h + 1] - m->h_hmmStateOff[h] != (N - 2) * NSt) {
         htkamd_set_error("model_create: HMM %d has %d emitting states but its transP has %d states", h,
                          (m->h_hmmStateOff[h + 1] - m->h_hmmStateOff[h]) / NSt, N);
         htkamd_model_destroy(m); return HTKAMD_EINVAL;
      }
   }
   int rc;
   if ((rc = toDevice(&m->d_stateCompOff, m->h_stateCompOff, (size_t)m->S + 1)) ||
       (rc = toDevice(&m->d_compGauss, m->h_compGauss, (size_t)m->C)) ||
       (rc = toDevice(&m->d_transN, m->h_transN, (size_t)m->nT)) ||
       (rc = toDevice(&m->d_transOff, m->h_transOff, (size_t)m->nT + 1)) ||
       (rc = model_refresh(m))) {
      htkamd_model_destroy(m); return rc;
   }
   if (d->hsKind == HTKAMD_HS_TIED) {
      m->tiedMix = 1; m->tmBeam = 10.0f;
      m->h_tmPoolOff = (int *)calloc((size_t)NSt + 1, sizeof(int));
      for (int k = 0; k < NSt; k++) m->h_tmPoolOff[k + 1] = m->h_tmPoolOff[k] + (m->h_stateCompOff[k + 1] - m->h_stateCompOff[k]);
      m->tmPool = m->h_tmPoolOff[NSt];
      for (int e = 0; e < m->S; e++) {
         const int k = e % NSt, c0 = m->h_stateCompOff[e], M = m->h_stateCompOff[e + 1] - c0, p0 = m->h_stateCompOff[k];
         bool same = M == m->h_stateCompOff[k + 1] - p0;
         for (int i = 0; same && i < M; i++) same = m->h_compGauss[c0 + i] == m->h_compGauss[p0 + i];
         if (!same) { htkamd_set_error("model_create: tied-mixture set: stream %d of state %d does not list its stream's pool", k + 1, e / NSt); htkamd_model_destroy(m); return HTKAMD_EINVAL; }
         if (M < 2) { htkamd_set_error("model_create: tied-mixture set: a pool of one Gaussian"); htkamd_model_destroy(m); return HTKAMD_EINVAL; }
      }
      if ((rc = toDevice(&m->d_tmPoolOff, m->h_tmPoolOff, (size_t)NSt + 1)) || (rc = htkamd_model_device_tables(m))) { htkamd_model_destroy(m); return rc; }
   }
   if (NSt > 1 || m->tiedMix) {
      std::vector<int> two((size_t)m->S + 1);
      for (int e = 0; e <= m->S; e++) two[e] = 2 * e;
      if ((NSt > 1 && ((rc = toDevice(&m->d_dimStream, m->h_dimStream, (size_t)m->D)) || (rc = toDevice(&m->d_gaussStream, m->h_gaussStream, (size_t)m->G)) ||
                       (rc = toDevice(&m->d_streamWt, m->h_streamWt, (size_t)m->S)))) ||
          (rc = toDevice(&m->d_msCompOff, two.data(), (size_t)m->S + 1))) { htkamd_model_destroy(m); return rc; }
   }
   {
      const int n = htkamd_host_ladd_table_size();
      double *tab = (double *)malloc(sizeof(double) * (size_t)n);
      htkamd_host_build_ladd_table(tab);
      rc = toDevice(&m->d_laddTab, tab, (size_t)n);
      free(tab);
      if (rc) { htkamd_model_destroy(m); return rc; }
   }
   *out = m;
   return HTKAMD_OK;
}

extern "C" void htkamd_model_destroy(htkamd_model *m)
{
   if (!m) return;
   free(m->h_stateCompOff); free(m->h_compGauss); free(m->h_transN); free(m->h_transOff); free(m->h_hmmTrans);
   free(m->h_hmmStateOff); free(m->h_hmmState); free(m->h_minDur); free(m->h_transLR); free(m->h_trOccOff); free(m->h_scanOrder);
   free(m->h_mean); free(m->h_var); free(m->h_ivar); free(m->h_gconst); free(m->h_compWeight); free(m->h_compLogWt); free(m->h_transP);
   (void)hipFree(m->d_gparam); (void)hipFree(m->d_laddTab); (void)hipFree(m->d_mean); (void)hipFree(m->d_ivar); (void)hipFree(m->d_gconst);
   (void)hipFree(m->d_compLogWt); (void)hipFree(m->d_transP); (void)hipFree(m->d_stateCompOff); (void)hipFree(m->d_compGauss);
   (void)hipFree(m->d_transN); (void)hipFree(m->d_transOff); (void)hipFree(m->d_mfmaTab); (void)hipFree(m->d_stateTileOff);
   (void)hipFree(m->d_bf16Tab); (void)hipFree(m->d_f16Tab); (void)hipFree(m->d_f16Ctl); (void)hipFree(m->d_tileState);
   (void)hipFree(m->d_var); (void)hipFree(m->d_compWeight); (void)hipFree(m->d_trOccOff); (void)hipFree(m->d_hmmTrans);
   (void)hipFree(m->d_hmmStateOff); (void)hipFree(m->d_hmmState); (void)hipFree(m->d_updScratch);
   htkamd_outp_ring_free(m->obRing);
   free(m->h_meanLeader); free(m->h_varLeader); free(m->h_varGroupSize); (void)hipFree(m->d_shareTab);
   if (m->h_updPin) (void)hipHostFree(m->h_updPin);
   if (m->evUpd) (void)hipEventDestroy((hipEvent_t)m->evUpd);
   free(m->h_rawLogWt); (void)hipFree(m->d_rawLogWt);
   free(m->h_tmPoolOff); (void)hipFree(m->d_tmPoolOff); free(m->h_streamWt); (void)hipFree(m->d_streamWt);
   free(m->h_dimStream); free(m->h_gaussStream); (void)hipFree(m->d_dimStream); (void)hipFree(m->d_gaussStream); (void)hipFree(m->d_msCompOff);
   free(m);
}

static int compat_raw_logwt(htkamd_model *m);
// Shared mean / variance vectors: share[g] >= 0 names the vector Gaussian g's mean (variance) is a copy of, -1 = private.
extern "C" int htkamd_model_set_scan_order(htkamd_model *m, const int *order)
{
   if (!m) { htkamd_set_error("model_set_scan_order: NULL"); return HTKAMD_EINVAL; }
   free(m->h_scanOrder); m->h_scanOrder = nullptr;
   if (!order) return HTKAMD_OK;
   unsigned char *seen = (unsigned char *)calloc((size_t)(m->H ? m->H : 1), 1);
   for (int k = 0; k < m->H; k++) {
      if (order[k] < 0 || order[k] >= m->H || seen[order[k]]) { free(seen); htkamd_set_error("model_set_scan_order: not a permutation of the %d physical models", m->H); return HTKAMD_EINVAL; }
      seen[order[k]] = 1;
   }
   free(seen);
   m->h_scanOrder = (int *)malloc(sizeof(int) * (size_t)(m->H ? m->H : 1));
   memcpy(m->h_scanOrder, order, sizeof(int) * (size_t)m->H);
   if (m->compat & HTKAMD_COMPAT_SHARED_LOGWT) {           // which state is the FIRST user of a shared pdf depends on the order
      int rc = compat_raw_logwt(m);
      if (rc) return rc;
      if ((rc = model_refresh(m))) return rc;
      m->bf16Stale = 1; m->f16Stale = 1;
   }
   return HTKAMD_OK;
}

extern "C" int htkamd_model_set_sharing(htkamd_model *m, const int *meanShare, const int *varShare)
{
   if (!m) { htkamd_set_error("model_set_sharing: NULL model"); return HTKAMD_EINVAL; }
   free(m->h_meanLeader); free(m->h_varLeader); free(m->h_varGroupSize);
   m->h_meanLeader = m->h_varLeader = m->h_varGroupSize = nullptr;
   if (m->d_shareTab) { (void)hipFree(m->d_shareTab); m->d_shareTab = nullptr; }
   bool any = false;
   for (int g = 0; g < m->G; g++) if ((meanShare && meanShare[g] >= 0) || (varShare && varShare[g] >= 0)) any = true;
   if (!any) return HTKAMD_OK;
   int rc = htkamd_model_sync_host(m);
   if (rc) return rc;
   const int G = m->G, D = m->D;
   m->h_meanLeader = (int *)malloc(sizeof(int) * (size_t)G);
   m->h_varLeader = (int *)malloc(sizeof(int) * (size_t)G);
   m->h_varGroupSize = (int *)calloc((size_t)G, sizeof(int));
   for (int pass = 0; pass < 2; pass++) {
      const int *share = pass ? varShare : meanShare;
      int *leader = pass ? m->h_varLeader : m->h_meanLeader;
      const float *vec = pass ? m->h_var : m->h_mean;
      int maxId = -1;
      for (int g = 0; g < G; g++) if (share && share[g] > maxId) maxId = share[g];
      std::vector<int> first((size_t)(maxId + 1), -1);
      for (int g = 0; g < G; g++) {
         leader[g] = g;
         if (!share || share[g] < 0) continue;
         if (first[share[g]] < 0) first[share[g]] = g;
         leader[g] = first[share[g]];
         if (memcmp(vec + (size_t)g * D, vec + (size_t)leader[g] * D, sizeof(float) * (size_t)D)) {
            htkamd_set_error("model_set_sharing: Gaussians %d and %d are declared to share a %s but hold different values", leader[g], g, pass ? "variance" : "mean");
            free(m->h_meanLeader); free(m->h_varLeader); free(m->h_varGroupSize);
            m->h_meanLeader = m->h_varLeader = m->h_varGroupSize = nullptr;
            return HTKAMD_EINVAL;
         }
      }
   }
   for (int g = 0; g < G; g++) m->h_varGroupSize[m->h_varLeader[g]]++;
   // the device update's copy: leaders, group sizes and, per leader, the other members of its group in ascending order (the order
   // in which htkamd_update_models pools their statistics)
   {
      std::vector<int> tab((size_t)3 * G + 2 * ((size_t)G + 1));
      memcpy(tab.data(), m->h_meanLeader, sizeof(int) * (size_t)G);
      memcpy(tab.data() + G, m->h_varLeader, sizeof(int) * (size_t)G);
      memcpy(tab.data() + 2 * (size_t)G, m->h_varGroupSize, sizeof(int) * (size_t)G);
      std::vector<int> mem[2];
      for (int pass = 0; pass < 2; pass++) {
         const int *leader = pass ? m->h_varLeader : m->h_meanLeader;
         int *off = tab.data() + 3 * (size_t)G + (size_t)pass * (G + 1);
         std::vector<int> cnt((size_t)G + 1, 0);
         for (int g = 0; g < G; g++) if (leader[g] != g) cnt[leader[g] + 1]++;
         off[0] = 0;
         for (int g = 0; g < G; g++) off[g + 1] = off[g] + cnt[g + 1];
         mem[pass].resize((size_t)off[G] + 1);
         std::vector<int> fill(off, off + G);
         for (int g = 0; g < G; g++) if (leader[g] != g) mem[pass][fill[leader[g]]++] = g;
         (pass ? m->shareVaMem : m->shareMuMem) = off[G];
      }
      tab.insert(tab.end(), mem[0].begin(), mem[0].begin() + m->shareMuMem);
      tab.insert(tab.end(), mem[1].begin(), mem[1].begin() + m->shareVaMem);
      HIPCHECK(hipMalloc(&m->d_shareTab, sizeof(int) * tab.size()));
      HIPCHECK(hipMemcpy(m->d_shareTab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
   }
   return HTKAMD_OK;
}

extern "C" int htkamd_model_set_tm_beam(htkamd_model *m, float tmBeam)
{
   if (!m || !(tmBeam >= 0.0f)) { htkamd_set_error("model_set_tm_beam: bad argument"); return HTKAMD_EINVAL; }
   m->tmBeam = tmBeam;
   return HTKAMD_OK;
}

// HTKAMD_COMPAT_SHARED_LOGWT: ConvLogWt (HUtil.c:474-485) walks the set with GoNextMix(noSkip = FALSE), which passes over a mixture
// component whose pdf it has met before (HUtil.c:371-394: IsSeen(mp->nUse)) -- the WEIGHT of that component stays linear and is read as
// a log weight from then on (MixLogWeight returns it as it stands once hset->logWt is set, HModel.c:5290).  The walk: models in HMM scan
// order, a model's states 2 .. N - 1 (a state met before is skipped whole), streams, components.
static int compat_raw_logwt(htkamd_model *m)
{
   free(m->h_rawLogWt); m->h_rawLogWt = nullptr;
   if (m->d_rawLogWt) { (void)hipFree(m->d_rawLogWt); m->d_rawLogWt = nullptr; }
   if (!(m->compat & HTKAMD_COMPAT_SHARED_LOGWT) || m->tiedMix) return HTKAMD_OK;
   unsigned char *raw = (unsigned char *)calloc((size_t)(m->C ? m->C : 1), 1);
   const int NSt = m->NSt > 1 ? m->NSt : 1;
   unsigned char *seenS = (unsigned char *)calloc((size_t)(m->S ? m->S : 1) * NSt, 1), *seenG = (unsigned char *)calloc((size_t)(m->G ? m->G : 1), 1);
   int any = 0;
   for (int hh = 0; hh < m->H; hh++) {
      const int h = m->h_scanOrder ? m->h_scanOrder[hh] : hh;
      for (int j = m->h_hmmStateOff[h]; j < m->h_hmmStateOff[h + 1]; j++) {
         const int e = m->h_hmmState[j];                    // a (state, stream) element: the list holds a state's streams one after the other
         if (seenS[e]) continue;                            // (a state met before is skipped whole: its elements with it)
         seenS[e] = 1;
         for (int c = m->h_stateCompOff[e]; c < m->h_stateCompOff[e + 1]; c++) {
            const int g = m->h_compGauss[c];
            if (seenG[g]) { raw[c] = 1; any = 1; } else seenG[g] = 1;
         }
      }
   }
   free(seenS); free(seenG);
   if (!any) { free(raw); return HTKAMD_OK; }             // no pdf is shared: nothing differs
   m->h_rawLogWt = raw;
   HIPCHECK(hipMalloc((void **)&m->d_rawLogWt, (size_t)m->C));
   HIPCHECK(hipMemcpy(m->d_rawLogWt, raw, (size_t)m->C, hipMemcpyHostToDevice));
   return HTKAMD_OK;
}

extern "C" int htkamd_model_set_compat(htkamd_model *m, int flags)
{
   if (!m || (flags & ~(HTKAMD_COMPAT_STREAM_REVISIT | HTKAMD_COMPAT_SHARED_LOGWT))) { htkamd_set_error("model_set_compat: unknown flags %d", flags); return HTKAMD_EINVAL; }
   const bool wtChange = ((m->compat ^ flags) & HTKAMD_COMPAT_SHARED_LOGWT) != 0;
   m->compat = flags;
   if (wtChange) {
      int rc = compat_raw_logwt(m);
      if (rc) return rc;
      if ((rc = model_refresh(m))) return rc;
      m->bf16Stale = 1; m->f16Stale = 1;
   }
   return HTKAMD_OK;
}

extern "C" int htkamd_model_has_sharing(const htkamd_model *m) { return m && m->h_meanLeader ? 1 : 0; }

extern "C" int htkamd_model_set_params(htkamd_model *m, const float *mean, const float *var, const float *gconst,
                                       const float *compWeight, const float *transP)
{
   if (!m) { htkamd_set_error("model_set_params: NULL model"); return HTKAMD_EINVAL; }
   { int rc0 = htkamd_model_sync_host(m); if (rc0) return rc0; }
   if (mean) memcpy(m->h_mean, mean, sizeof(float) * (size_t)m->G * m->D);
   if (var) memcpy(m->h_var, var, sizeof(float) * (size_t)m->G * m->D);
   if (gconst) memcpy(m->h_gconst, gconst, sizeof(float) * (size_t)m->G);
   else if (var)
      for (int g = 0; g < m->G; g++) htkamd_host_fix_diag_gconst_ms(m->D, m->h_var + (size_t)g * m->D, m->h_dimStream, m->h_gaussStream ? m->h_gaussStream[g] : 0, m->h_gconst + g);
   if (compWeight) memcpy(m->h_compWeight, compWeight, sizeof(float) * (size_t)m->C);
   if (transP) memcpy(m->h_transP, transP, sizeof(float) * (size_t)m->h_transOff[m->nT]);
   return model_refresh(m);
}

// Prepared tables given by the caller (an HTKLib front-end that has already run ConvDiagC / ConvLogWt / FixGConsts on its HMMSet):
// the kernels then read the caller's own floats, bit for bit, instead of values re-derived from variances and linear weights.
extern "C" int htkamd_model_set_prepared(htkamd_model *m, const float *ivar, const float *gconst, const float *compLogWt)
{
   if (!m) { htkamd_set_error("model_set_prepared: NULL model"); return HTKAMD_EINVAL; }
   { int rc0 = htkamd_model_sync_host(m); if (rc0) return rc0; }
   if (ivar) memcpy(m->h_ivar, ivar, sizeof(float) * (size_t)m->G * m->D);
   if (gconst) memcpy(m->h_gconst, gconst, sizeof(float) * (size_t)m->G);
   if (compLogWt) memcpy(m->h_compLogWt, compLogWt, sizeof(float) * (size_t)m->C);
   return model_refresh(m, false);
}

extern "C" int htkamd_model_get_prepared(htkamd_model *m, float *ivar, float *gconst, float *compLogWt, int *minDur)
{
   if (!m) { htkamd_set_error("model_get_prepared: NULL model"); return HTKAMD_EINVAL; }
   // read back from the DEVICE copies so that tests see what the kernels see
   if (ivar) HIPCHECK(hipMemcpy(ivar, m->d_ivar, sizeof(float) * (size_t)m->G * m->D, hipMemcpyDeviceToHost));
   if (gconst) HIPCHECK(hipMemcpy(gconst, m->d_gconst, sizeof(float) * (size_t)m->G, hipMemcpyDeviceToHost));
   if (compLogWt) HIPCHECK(hipMemcpy(compLogWt, m->d_compLogWt, sizeof(float) * (size_t)m->C, hipMemcpyDeviceToHost));
   if (minDur) memcpy(minDur, m->h_minDur, sizeof(int) * (size_t)m->nT);
   return HTKAMD_OK;
}

extern "C" int htkamd_model_update(htkamd_model *m, const htkamd_accs *accs, const double *hostVec,
                                   const htkamd_update_config *cfg, htkamd_update_stats *stats)
{
   if (!m || !accs || !hostVec || !cfg || !stats) { htkamd_set_error("model_update: NULL argument"); return HTKAMD_EINVAL; }
   if (accs->m != m) { htkamd_set_error("model_update: accumulators belong to a different model"); return HTKAMD_EINVAL; }
   int rc = htkamd_model_sync_host(m);
   if (rc) return rc;
   rc = htkamd_update_models(m, &accs->lay, hostVec, cfg, stats);
   if (rc && rc != HTKAMD_EMODEL) return rc;
   const int rc2 = model_refresh(m);
   return rc2 ? rc2 : rc;
}

extern "C" int htkamd_model_get_params(htkamd_model *m, float *mean, float *var, float *gconst, float *compWeight, float *transP)
{
   if (!m) { htkamd_set_error("model_get_params: NULL model"); return HTKAMD_EINVAL; }
   { int rc0 = htkamd_model_sync_host(m); if (rc0) return rc0; }
   if (mean) memcpy(mean, m->h_mean, sizeof(float) * (size_t)m->G * m->D);
   if (var) memcpy(var, m->h_var, sizeof(float) * (size_t)m->G * m->D);
   if (gconst) memcpy(gconst, m->h_gconst, sizeof(float) * (size_t)m->G);
   if (compWeight) memcpy(compWeight, m->h_compWeight, sizeof(float) * (size_t)m->C);
   if (transP) memcpy(transP, m->h_transP, sizeof(float) * (size_t)m->h_transOff[m->nT]);
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ accumulators

extern "C" int htkamd_accs_create(htkamd_model *m, htkamd_accs **out)
{
   if (!m || !out) { htkamd_set_error("accs_create: NULL argument"); return HTKAMD_EINVAL; }
   htkamd_accs *a = (htkamd_accs *)calloc(1, sizeof(htkamd_accs));
   a->m = m;
   size_t o = 0, GD = (size_t)m->G * m->D;
   a->lay.mu = o; o += GD;
   a->lay.muOcc = o; o += m->G;
   a->lay.va = o; o += GD;
   a->lay.vaOcc = o; o += m->G;
   a->lay.wt = o; o += m->C;
   a->lay.wtOcc = o; o += m->S;
   a->lay.tr = o; o += m->h_transOff[m->nT];
   a->lay.trOcc = o; o += m->h_trOccOff[m->nT];
   a->lay.nEgs = o; o += m->H;
   a->lay.totalPr = o++; a->lay.totalT = o++; a->lay.nUttDone = o++; a->lay.nUttSkipped = o++; a->lay.nEval = o++;
   a->lay.total = o;
   // (room to the next 256 bytes behind the vector: htkamd_accs_zero clears whole 256-byte blocks -- a size that is not such a multiple is
   //  filled by two kernels, bulk and tail, 4 us apart in every EM iteration)
   const size_t oPad = (o + 31) & ~(size_t)31;
   hipError_t e = hipMalloc((void **)&a->d_vec, sizeof(double) * oPad);
   if (e != hipSuccess) { htkamd_set_error("accs_create: hipMalloc(%zu doubles): %s", o, hipGetErrorString(e)); free(a); return HTKAMD_ENOMEM; }
   e = hipMemset(a->d_vec, 0, sizeof(double) * oPad);
   if (e != hipSuccess) { htkamd_set_error("accs_create: hipMemset: %s", hipGetErrorString(e)); (void)hipFree(a->d_vec); free(a); return HTKAMD_EHIP; }
   *out = a;
   return HTKAMD_OK;
}

extern "C" void htkamd_accs_destroy(htkamd_accs *a)
{
   if (!a) return;
   (void)hipFree(a->d_vec);
   free(a);
}

extern "C" int htkamd_accs_zero(htkamd_accs *a, void *stream)
{
   if (!a) { htkamd_set_error("accs_zero: NULL"); return HTKAMD_EINVAL; }
   HIPCHECK(hipMemsetAsync(a->d_vec, 0, sizeof(double) * ((a->lay.total + 31) & ~(size_t)31), (hipStream_t)stream));
   return HTKAMD_OK;
}

extern "C" int htkamd_accs_get_layout(const htkamd_accs *a, htkamd_accs_layout *out)
{
   if (!a || !out) { htkamd_set_error("accs_get_layout: NULL"); return HTKAMD_EINVAL; }
   *out = a->lay;
   return HTKAMD_OK;
}

extern "C" int htkamd_accs_device_vector(htkamd_accs *a, double **dVec, size_t *n)
{
   if (!a || !dVec || !n) { htkamd_set_error("accs_device_vector: NULL"); return HTKAMD_EINVAL; }
   *dVec = a->d_vec; *n = a->lay.total;
   return HTKAMD_OK;
}

extern "C" int htkamd_accs_download(htkamd_accs *a, double *hostVec, void *stream)
{
   if (!a || !hostVec) { htkamd_set_error("accs_download: NULL"); return HTKAMD_EINVAL; }
   HIPCHECK(hipMemcpyAsync(hostVec, a->d_vec, sizeof(double) * a->lay.total, hipMemcpyDeviceToHost, (hipStream_t)stream));
   HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
   return HTKAMD_OK;
}

__global__ void k_vec_add(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
   size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
   for (; i < n; i += st) dst[i] += src[i];
}

extern "C" int htkamd_accs_upload_add(htkamd_accs *a, const double *hostVec, void *stream)
{
   if (!a || !hostVec) { htkamd_set_error("accs_upload_add: NULL"); return HTKAMD_EINVAL; }
   double *tmp = nullptr;
   HIPCHECK(hipMalloc((void **)&tmp, sizeof(double) * a->lay.total));
   struct Free { double *p; ~Free() { (void)hipFree(p); } } freeTmp{tmp};      // released on every path
   hipStream_t s = (hipStream_t)stream;
   HIPCHECK(hipMemcpyAsync(tmp, hostVec, sizeof(double) * a->lay.total, hipMemcpyHostToDevice, s));
   hipLaunchKernelGGL(k_vec_add, dim3(1024), dim3(256), 0, s, a->d_vec, tmp, a->lay.total);
   HIPCHECK(hipGetLastError());
   HIPCHECK(hipStreamSynchronize(s));
   return HTKAMD_OK;
}
