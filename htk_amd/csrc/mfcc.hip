// mfcc.hip -- K6: waveform -> MFCC(+_0/_E)(+_D)(+_A)(+_Z) on the device (HSigP/HParm front end).
//
// The reference converts one frame at a time with float data, double-precision twiddle recurrences and libm
// calls (HParm.c:2214 ConvertFrame; HSigP.c).  Everything libm-dependent that does not depend on the signal
// (Hamming window, mel weights, lifter, DCT cosines, FFT twiddle recurrences) is tabulated on the host exactly as
// the reference computes it (host/fbank.c); the kernels then perform the reference's multiply/add sequences in
// the reference's order, so the output equals HCopy's up to the device's double log()/sqrt() rounding.
//
// MI355X mapping: frames are independent, so one wavefront owns one frame (thousands of frames in flight).  The
// 512-point real FFT is a 256-point complex radix-2 DIT in LDS (2 butterflies per lane and stage, double
// arithmetic, float storage -- as the reference); mel bins and cepstra are sequential float sums in the
// reference, so a lane owns a bin / a cepstral coefficient and walks its k range in order.  Frame energies and
// the source mean are 400-term sequential float sums: a lane per FRAME does those in a separate tiny kernel.
// Deltas/accelerations are one thread per output element; energy normalisation and _Z are per-utterance passes.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "internal.h"
#include "hipcheck.h"

struct MfccArgs {
   const short *wav;
   const long long *frameSamp;   // [F] first sample of each frame
   int nFrames;
   int frSize, fftN, klo, khi, numChans, numCeps, nCols, nStat;
   float preEmph, cepScale, mfnorm;
   int useHam, usePower, zMean, hasC0, hasE, rawEnergy;
   const float *ham, *cepWin, *loWt;
   const int *binA0, *binA1, *binB0, *binB1;
   const double *dct, *tw, *rtw;
   const short *brev;
   float *frameMean;             // [F] (ZMEANSOURCE)
   float *out;
};

// first sample and utterance of every frame from the per-utterance offsets (binary search over the utterances)
__global__ void k_mfcc_index(const int *frameOff, const int *sampOff, int nUtt, int nFrames, int frRate, long long *frameSamp, int *frameUtt)
{
   const int f = blockIdx.x * blockDim.x + threadIdx.x;
   if (f >= nFrames) return;
   int lo = 0, hi = nUtt - 1;                            // the last u with frameOff[u] <= f
   while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (frameOff[mid] <= f) lo = mid; else hi = mid - 1; }
   frameUtt[f] = lo;
   frameSamp[f] = (long long)sampOff[lo] + (long long)(f - frameOff[lo]) * frRate;
}

// lane per frame: source mean (ZeroMeanFrame HParm.c:2132) and log energy (HParm.c:2234-2238 / HSigP.c:571-575)
__global__ void k_mfcc_energy(MfccArgs a)
{
   const int f = blockIdx.x * blockDim.x + threadIdx.x;
   if (f >= a.nFrames) return;
   const short *w = a.wav + a.frameSamp[f];
   float off = 0.0f;
   if (a.zMean) {
      float sum = 0.0f;
      for (int i = 0; i < a.frSize; i++) sum += (float)w[i];
      off = sum / a.frSize;
      a.frameMean[f] = off;
   }
   if (!a.hasE) return;
   float te = 0.0f;
   if (a.rawEnergy) {
      for (int i = 0; i < a.frSize; i++) { const float s = (float)w[i] - off; te += s * s; }
   } else {                                              // energy of the pre-emphasised, windowed frame
      float prev = 0.0f;
      for (int i = 0; i < a.frSize; i++) {
         const float cur = (float)w[i] - off;
         float s;
         if (a.preEmph > 0.0f) s = (i == 0) ? (float)((double)cur * (1.0 - (double)a.preEmph)) : cur - prev * a.preEmph;
         else s = cur;
         if (a.useHam) s *= a.ham[i + 1];
         te += (s * s);
         prev = cur;
      }
   }
   a.out[(size_t)f * a.nCols + a.nStat - 1] = (te < MINLARG) ? (float)LZERO : (float)log((double)te);
}

// One frame's spectrum by a whole wavefront: window, FFT, Realft, magnitudes -- and (round 6) what every k hands its two mel bins:
//   uk[k] = ek - loWt[k] ek  (to the bin above, HSigP.c:594)   vk[k] = loWt[k] ek  (to its own bin, :593)
// computed by all lanes in parallel, so that the bins' sums below are nothing but the reference's chain of float additions.
__device__ __forceinline__ void mfcc_spectrum(const MfccArgs &a, const int f, float *xs, float *uk, float *vk, const int lane)
{
   const int fftN = a.fftN, nn = fftN / 2;
   const short *w = a.wav + a.frameSamp[f];
   const float off = a.zMean ? a.frameMean[f] : 0.0f;

   // ---- load, pre-emphasise (HSigP.c:134), window (HSigP.c:122), zero-pad, bit-reverse the complex index
#ifndef MFCC_BREV_READ
#define MFCC_BREV_READ 0                                /* 1: the reversal on the read side, LDS written in order -- measured 6 % SLOWER (tools/r06_mfcc2.sh): the index load sits in front of the samples' loads */
#endif
   for (int r_ = lane; r_ < nn; r_ += 64) {
      const int c = MFCC_BREV_READ ? (int)a.brev[r_] : r_, r = MFCC_BREV_READ ? r_ : (int)a.brev[r_];
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
         const int i = 2 * c + h;
         float s = 0.0f;
         if (i < a.frSize) {
            const float cur = (float)w[i] - off;
            if (a.preEmph > 0.0f) {
               if (i == 0) s = (float)((double)cur * (1.0 - (double)a.preEmph));
               else { const float prev = (float)w[i - 1] - off; s = cur - prev * a.preEmph; }
            } else s = cur;
            if (a.useHam) s *= a.ham[i + 1];
         }
         v[h] = s;
      }
      *(float2 *)(xs + 2 * r) = make_float2(v[0], v[1]);
   }
   __syncthreads();
   // ---- complex FFT, radix-2 DIT (HSigP.c:332-349): stage with half-size h, twiddles tabulated
   {
      // Stages in PAIRS (round 6): a lane takes the four elements base + {0, h, 2h, 3h} through the stages of half-size h and 2h in registers --
      // the same butterflies on the same floats (every result is rounded to float where the reference stores it), so the output is unchanged
      // bit for bit, with half the passes through LDS and half the barriers.  Twiddles of the stage of half-size h start at h - 1 in the table.
      auto bfly = [](float &ar, float &ai, float &br, float &bi, const double wr, const double wi) {
         const double xre = wr * (double)br - wi * (double)bi;
         const double xri = wr * (double)bi + wi * (double)br;
         const float nbr = (float)((double)ar - xre), nbi = (float)((double)ai - xri);
         const float nar = (float)((double)ar + xre), nai = (float)((double)ai + xri);
         ar = nar; ai = nai; br = nbr; bi = nbi;
      };
      int h = 1, lg = 0;
      for (; 4 * h <= nn; h *= 4, lg += 2) {
         for (int g = lane; g < nn / 4; g += 64) {
            const int pos = g & (h - 1), base = ((g >> lg) << (lg + 2)) + pos;
            float2 e0, e1, e2, e3;
            if (h == 1) {                                 // four neighbours: two 16-byte words, the lanes' in order (8-byte reads 32 bytes apart fell into four banks)
               const float4 lo4 = *(const float4 *)(xs + 2 * base), hi4 = *(const float4 *)(xs + 2 * base + 4);
               e0 = make_float2(lo4.x, lo4.y); e1 = make_float2(lo4.z, lo4.w); e2 = make_float2(hi4.x, hi4.y); e3 = make_float2(hi4.z, hi4.w);
            } else {
               e0 = *(const float2 *)(xs + 2 * base); e1 = *(const float2 *)(xs + 2 * (base + h));
               e2 = *(const float2 *)(xs + 2 * (base + 2 * h)); e3 = *(const float2 *)(xs + 2 * (base + 3 * h));
            }
            const double2 t1 = *(const double2 *)(a.tw + 2 * (h - 1 + pos));
            const double2 t2 = *(const double2 *)(a.tw + 2 * (2 * h - 1 + pos)), t3 = *(const double2 *)(a.tw + 2 * (2 * h - 1 + pos + h));
            bfly(e0.x, e0.y, e1.x, e1.y, t1.x, t1.y); bfly(e2.x, e2.y, e3.x, e3.y, t1.x, t1.y);
            bfly(e0.x, e0.y, e2.x, e2.y, t2.x, t2.y); bfly(e1.x, e1.y, e3.x, e3.y, t3.x, t3.y);
            if (h == 1) {
               *(float4 *)(xs + 2 * base) = make_float4(e0.x, e0.y, e1.x, e1.y); *(float4 *)(xs + 2 * base + 4) = make_float4(e2.x, e2.y, e3.x, e3.y);
            } else {
               *(float2 *)(xs + 2 * base) = e0; *(float2 *)(xs + 2 * (base + h)) = e1;
               *(float2 *)(xs + 2 * (base + 2 * h)) = e2; *(float2 *)(xs + 2 * (base + 3 * h)) = e3;
            }
         }
         __syncthreads();
      }
      int twOff = h - 1;
      for (; h < nn; h *= 2) {                           // (an odd number of stages: the last one alone)
         for (int b = lane; b < nn / 2; b += 64) {
            const int grp = b / h, pos = b % h;
            const int ia = grp * 2 * h + pos, ib = ia + h;
            const double wr = a.tw[2 * (twOff + pos)], wi = a.tw[2 * (twOff + pos) + 1];
            const float sjr = xs[2 * ib], sji = xs[2 * ib + 1], sir = xs[2 * ia], sii = xs[2 * ia + 1];
            const double xre = wr * (double)sjr - wi * (double)sji;
            const double xri = wr * (double)sji + wi * (double)sjr;
            xs[2 * ib] = (float)((double)sir - xre); xs[2 * ib + 1] = (float)((double)sii - xri);
            xs[2 * ia] = (float)((double)sir + xre); xs[2 * ia + 1] = (float)((double)sii + xri);
         }
         twOff += h;
         __syncthreads();
      }
   }
   // ---- Realft post-pass (HSigP.c:371-390), 1-based indices of the reference mapped to xs[idx-1]
   {
      const int n = nn, n2 = n / 2;
      for (int i = 2 + lane; i <= n2; i += 64) {
         const int i1 = i + i - 1, i2 = i1 + 1, i3 = n + n + 3 - i2, i4 = i3 + 1;
         const double wrs = a.rtw[2 * i], wis = a.rtw[2 * i + 1];
         const float s1 = xs[i1 - 1], s2 = xs[i2 - 1], s3 = xs[i3 - 1], s4 = xs[i4 - 1];
         const double xr1 = ((double)(s1 + s3)) / 2.0, xi1 = ((double)(s2 - s4)) / 2.0;
         const double xr2 = ((double)(s2 + s4)) / 2.0, xi2 = ((double)(s3 - s1)) / 2.0;
         xs[i1 - 1] = (float)(xr1 + wrs * xr2 - wis * xi2);
         xs[i2 - 1] = (float)(xi1 + wrs * xi2 + wis * xr2);
         xs[i3 - 1] = (float)(xr1 - wrs * xr2 + wis * xi2);
         xs[i4 - 1] = (float)(-xi1 + wrs * xi2 + wis * xr2);
      }
      __syncthreads();
      if (lane == 0) { const float xr1 = xs[0]; xs[0] = xr1 + xs[1]; xs[1] = 0.0f; }
      __syncthreads();
   }
   // ---- magnitudes (HSigP.c:585-590) and the two terms of every k (:593-594)
   for (int k = a.klo + lane; k <= a.khi; k += 64) {
      const float t1 = xs[2 * k - 2], t2 = xs[2 * k - 1];
      const float p = t1 * t1 + t2 * t2;
      const float e = a.usePower ? p : (float)sqrt((double)p);
      const float tw = a.loWt[k] * e;
      uk[k] = e - tw; vk[k] = tw;
   }
   __syncthreads();
}

// mel bin b of one frame (HSigP.c:591-594 in the reference's accumulation order, then log with floor 1.0, :598-603)
__device__ __forceinline__ float mfcc_bin(const MfccArgs &a, const int b, const float *uk, const float *vk)
{
   float acc = 0.0f;
   for (int k = a.binA0[b]; k <= a.binA1[b]; k++) acc += uk[k];
   for (int k = a.binB0[b]; k <= a.binB1[b]; k++) acc += vk[k];
   if (acc < 1.0f) acc = 1.0f;
   return (float)log((double)acc);
}
// cepstral coefficient j of one frame: DCT (HSigP.c:607-621), lifter (:773), CEPSCALE
__device__ __forceinline__ float mfcc_cep(const MfccArgs &a, const int j, const float *fb)
{
   float c = 0.0f;
   const double *ct = a.dct + (size_t)j * (a.numChans + 1);
   for (int k = 1; k <= a.numChans; k++) c = (float)((double)c + (double)fb[k] * ct[k]);
   c *= a.mfnorm;
   c *= a.cepWin[j];
   return c * a.cepScale;
}
__device__ __forceinline__ float mfcc_c0(const MfccArgs &a, const float *fb)      // C0 (HSigP.c:647)
{
   float sum = 0.0f;
   for (int k = 1; k <= a.numChans; k++) sum += fb[k];
   return (sum * a.mfnorm) * a.cepScale;
}

// One wavefront per PAIR of frames (round 6).  The spectra take the whole wavefront, one frame after the other; the mel bins and the cepstra
// are chains of dependent float additions that only numChans (26) and numCeps + 1 (13) lanes can work on -- 40 % of the kernel's vector
// instructions ran with 26 lanes on -- so the two frames' chains run side by side, frame h in lanes 32 h .. 32 h + 31.
// PAIR = false (numChans > 32 or numCeps > 31): one frame per wavefront, the bins and cepstra strided over the lanes.
template <bool PAIR>
__global__ __launch_bounds__(64) void k_mfcc_frames(MfccArgs a)
{
   extern __shared__ float lds[];
   const int lane = threadIdx.x;
   const int fftN = a.fftN, nn = fftN / 2;
   constexpr int NF = PAIR ? 2 : 1;
   const int per = (fftN + 2 * (nn + 2) + a.numChans + 2 + 3) & ~3;      // floats per frame (a multiple of 16 bytes): xs [fftN] interleaved (re, im) | uk [nn + 2] | vk [nn + 2], 1-based k | fb [numChans + 2]
   for (int blk = blockIdx.x; NF * blk < a.nFrames; blk += gridDim.x) {
   const int f0 = NF * blk;
   for (int h = 0; h < NF; h++)
      if (f0 + h < a.nFrames) mfcc_spectrum(a, f0 + h, lds + h * per, lds + h * per + fftN, lds + h * per + fftN + nn + 2, lane);
   if constexpr (PAIR) {
      const int h = lane >> 5, q = lane & 31, f = f0 + h;
      float *base = lds + h * per, *fb = base + fftN + 2 * (nn + 2);
      if (f < a.nFrames && q < a.numChans) fb[q + 1] = mfcc_bin(a, q + 1, base + fftN, base + fftN + nn + 2);
      __syncthreads();
      if (f < a.nFrames) {
         float *row = a.out + (size_t)f * a.nCols;
         if (q < a.numCeps) row[q] = mfcc_cep(a, q + 1, fb);
         else if (a.hasC0 && q == 31) row[a.numCeps] = mfcc_c0(a, fb);
      }
   } else {
      float *fb = lds + fftN + 2 * (nn + 2);
      for (int b = 1 + lane; b <= a.numChans; b += 64) fb[b] = mfcc_bin(a, b, lds + fftN, lds + fftN + nn + 2);
      __syncthreads();
      float *row = a.out + (size_t)f0 * a.nCols;
      for (int j = 1 + lane; j <= a.numCeps; j += 64) row[j - 1] = mfcc_cep(a, j, fb);
      if (a.hasC0 && lane == 63) row[a.numCeps] = mfcc_c0(a, fb);
   }
   __syncthreads();                                      // (the next frames' spectra overwrite what the cepstra read)
   }
}

// NormaliseLogEnergy (HSigP.c:911-932): one block per utterance
__global__ void k_mfcc_enorm(float *out, const int *frameOff, int nCols, int col, float silFloor, float eScale)
{
   __shared__ float red[256];
   const int u = blockIdx.x, f0 = frameOff[u], f1 = frameOff[u + 1];
   if (f1 <= f0) return;
   float mx = out[(size_t)f0 * nCols + col];
   for (int f = f0 + threadIdx.x; f < f1; f += blockDim.x) mx = fmaxf(mx, out[(size_t)f * nCols + col]);
   red[threadIdx.x] = mx;
   __syncthreads();
   for (int o = blockDim.x / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
      __syncthreads();
   }
   mx = red[0];
   const float mn = (float)((double)mx - ((double)silFloor * log(10.0)) / 10.0);
   for (int f = f0 + threadIdx.x; f < f1; f += blockDim.x) {
      float p = out[(size_t)f * nCols + col];
      if (p < mn) p = mn;
      out[(size_t)f * nCols + col] = (float)(1.0 - (double)((mx - p) * eScale));
   }
}

// Regress (HSigP.c:827-856) on a whole table: out[t][ti+k] = sum_tau tau*(c[min(t+tau,last)] - c[max(t-tau,first)]) / (2 sum tau^2)
// mode bit 0: SIMPLEDIFFS = (c[t+w] - c[t-w]) / 2w (HSigP.c:848); bit 1: V1COMPAT = the first / last w rows of a table longer than 2w
// rows are forward / backward differences (AddHeadRegress / AddTailRegress with delwin 0, HSigP.c:866-908)
__global__ void k_mfcc_delta(float *out, const int *frameUtt, const int *frameOff, int nFrames, int nCols, int si, int ti, int d, int win, int mode = 0)
{
   const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= (size_t)nFrames * d) return;
   const int f = (int)(g / d), k = (int)(g % d);
   const int u = frameUtt[f], f0 = frameOff[u], f1 = frameOff[u + 1] - 1;
   if ((mode & 2) && (f1 - f0 + 1) - 2 * win > 0) {
      if (f - f0 < win) { out[(size_t)f * nCols + ti + k] = out[(size_t)(f + 1) * nCols + si + k] - out[(size_t)f * nCols + si + k]; return; }
      if (f1 - f < win) { out[(size_t)f * nCols + ti + k] = out[(size_t)f * nCols + si + k] - out[(size_t)(f - 1) * nCols + si + k]; return; }
   }
   if (mode & 1) {
      const int fb = (f - win < f0) ? f0 : f - win, ff = (f + win > f1) ? f1 : f + win;
      out[(size_t)f * nCols + ti + k] = (out[(size_t)ff * nCols + si + k] - out[(size_t)fb * nCols + si + k]) / (2 * win);
      return;
   }
   float sigmaT2 = 0.0f;
   for (int t = 1; t <= win; t++) sigmaT2 += t * t;
   sigmaT2 *= 2.0;
   float sum = 0.0f;
   for (int t = 1; t <= win; t++) {
      const int fb = (f - t < f0) ? f0 : f - t, ff = (f + t > f1) ? f1 : f + t;
      sum += t * (out[(size_t)ff * nCols + si + k] - out[(size_t)fb * nCols + si + k]);
   }
   out[(size_t)f * nCols + ti + k] = sum / sigmaT2;
}

// FZeroMean (HSigP.c:803-823): one thread per (utterance, column), double sum in frame order
__global__ void k_mfcc_zmean(float *out, const int *frameOff, int nUtt, int nCols, int d)
{
   const int g = blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= nUtt * d) return;
   const int u = g / d, k = g % d, f0 = frameOff[u], f1 = frameOff[u + 1];
   if (f1 <= f0) return;
   double sum = 0.0;
   for (int f = f0; f < f1; f++) sum += (double)out[(size_t)f * nCols + k];
   const float mean = (float)(sum / (double)(f1 - f0));
   for (int f = f0; f < f1; f++) out[(size_t)f * nCols + k] -= mean;
}

// ------------------------------------------------------------------------------------ host side
struct htkamd_mfcc {
   htkamd_mfcc_config cfg;
   htkamd_mfcc_tables tab;
   float *d_ham, *d_cepWin, *d_loWt, *d_frameMean;
   int *d_bins, *d_frameOff, *d_frameUtt;
   double *d_dct, *d_tw, *d_rtw;
   short *d_brev;
   long long *d_frameSamp;
   size_t capFrames, capUtt;
};

template <typename T> static int up(T **d, const T *h, size_t n)
{
   HIPCHECK(hipMalloc((void **)d, sizeof(T) * (n ? n : 1)));
   if (n) HIPCHECK(hipMemcpy(*d, h, sizeof(T) * n, hipMemcpyHostToDevice));
   return HTKAMD_OK;
}

extern "C" int htkamd_mfcc_create(const htkamd_mfcc_config *cfg, htkamd_mfcc **out)
{
   if (!cfg || !out) { htkamd_set_error("mfcc_create: NULL argument"); return HTKAMD_EINVAL; }
   if (htkamd_device_count() <= 0) { htkamd_set_error("mfcc_create: no HIP device"); return HTKAMD_ENODEV; }
   htkamd_mfcc *f = (htkamd_mfcc *)calloc(1, sizeof(htkamd_mfcc));
   f->cfg = *cfg;
   int rc = htkamd_mfcc_tables_build(cfg, &f->tab);
   if (rc) { free(f); return rc; }
   const htkamd_mfcc_tables &t = f->tab;
   const int nn = t.fftN / 2;
   if ((rc = up(&f->d_ham, t.ham, (size_t)t.frSize + 1)) || (rc = up(&f->d_cepWin, t.cepWin, (size_t)cfg->numCeps + 1)) ||
       (rc = up(&f->d_loWt, t.loWt, (size_t)nn + 2)) || (rc = up(&f->d_bins, t.binA0, (size_t)4 * (cfg->numChans + 2))) ||
       (rc = up(&f->d_dct, t.dct, (size_t)(cfg->numCeps + 1) * (cfg->numChans + 1))) || (rc = up(&f->d_tw, t.tw, (size_t)2 * nn)) ||
       (rc = up(&f->d_rtw, t.rtw, (size_t)2 * (nn / 2 + 2))) || (rc = up(&f->d_brev, t.brev, (size_t)nn))) {
      htkamd_mfcc_destroy(f); return rc;
   }
   *out = f;
   return HTKAMD_OK;
}

extern "C" void htkamd_mfcc_destroy(htkamd_mfcc *f)
{
   if (!f) return;
   (void)hipFree(f->d_ham); (void)hipFree(f->d_cepWin); (void)hipFree(f->d_loWt); (void)hipFree(f->d_bins); (void)hipFree(f->d_dct);
   (void)hipFree(f->d_tw); (void)hipFree(f->d_rtw); (void)hipFree(f->d_brev); (void)hipFree(f->d_frameSamp); (void)hipFree(f->d_frameMean);
   (void)hipFree(f->d_frameOff); (void)hipFree(f->d_frameUtt);
   htkamd_mfcc_tables_free(&f->tab);
   free(f);
}

extern "C" int htkamd_mfcc_compute(htkamd_mfcc *f, const short *dWav, const int *sampOff, int nUtt, int *frameOff, float *dOut, void *stream)
{
   if (!f || !sampOff || !frameOff || nUtt < 0 || (nUtt > 0 && (!dWav || !dOut))) { htkamd_set_error("mfcc_compute: bad argument"); return HTKAMD_EINVAL; }
   hipStream_t s = (hipStream_t)stream;
   const htkamd_mfcc_config &c = f->cfg;
   const htkamd_mfcc_tables &t = f->tab;
   const int nCols = htkamd_mfcc_num_cols(&c), nStat = c.numCeps + (c.hasC0 ? 1 : 0) + (c.hasE ? 1 : 0);
   frameOff[0] = 0;
   for (int u = 0; u < nUtt; u++) frameOff[u + 1] = frameOff[u] + htkamd_mfcc_num_frames(&c, sampOff[u + 1] - sampOff[u]);
   const int F = frameOff[nUtt];
   if (F == 0) return HTKAMD_OK;
   if ((size_t)F > f->capFrames) {
      (void)hipFree(f->d_frameSamp); (void)hipFree(f->d_frameMean); (void)hipFree(f->d_frameUtt);
      f->d_frameSamp = nullptr; f->d_frameMean = nullptr; f->d_frameUtt = nullptr;
      const size_t cap = (size_t)F + F / 8;
      HIPCHECK(hipMalloc((void **)&f->d_frameSamp, sizeof(long long) * cap));
      HIPCHECK(hipMalloc((void **)&f->d_frameMean, sizeof(float) * cap));
      HIPCHECK(hipMalloc((void **)&f->d_frameUtt, sizeof(int) * cap));
      f->capFrames = cap;
   }
   if ((size_t)nUtt + 1 > f->capUtt) {
      (void)hipFree(f->d_frameOff); f->d_frameOff = nullptr;
      HIPCHECK(hipMalloc((void **)&f->d_frameOff, sizeof(int) * 2 * ((size_t)nUtt + 1)));      // frame offsets, then sample offsets
      f->capUtt = (size_t)nUtt + 1;
   }
   // the per-frame tables (first sample, utterance) are made ON the device from the two per-utterance tables (round 6: 596 000 push_backs and
   // 7 MB of pageable copies were 0.7 ms of a 2.7 ms call)
   HIPCHECK(hipMemcpyAsync(f->d_frameOff, frameOff, sizeof(int) * ((size_t)nUtt + 1), hipMemcpyHostToDevice, s));
   HIPCHECK(hipMemcpyAsync(f->d_frameOff + f->capUtt, sampOff, sizeof(int) * ((size_t)nUtt + 1), hipMemcpyHostToDevice, s));
   hipLaunchKernelGGL(k_mfcc_index, dim3((F + 255) / 256), dim3(256), 0, s, f->d_frameOff, f->d_frameOff + f->capUtt, nUtt, F, t.frRate, f->d_frameSamp, f->d_frameUtt);
   HIPCHECK(hipGetLastError());

   MfccArgs a;
   a.wav = dWav; a.frameSamp = f->d_frameSamp; a.nFrames = F;
   a.frSize = t.frSize; a.fftN = t.fftN; a.klo = t.klo; a.khi = t.khi; a.numChans = c.numChans; a.numCeps = c.numCeps;
   a.nCols = nCols; a.nStat = nStat; a.preEmph = c.preEmph; a.cepScale = c.cepScale; a.mfnorm = t.mfnorm;
   a.useHam = c.useHam; a.usePower = c.usePower; a.zMean = c.zMeanSource; a.hasC0 = c.hasC0; a.hasE = c.hasE; a.rawEnergy = c.rawEnergy;
   a.ham = f->d_ham; a.cepWin = f->d_cepWin; a.loWt = f->d_loWt;
   a.binA0 = f->d_bins; a.binA1 = f->d_bins + (c.numChans + 2); a.binB0 = a.binA1 + (c.numChans + 2); a.binB1 = a.binB0 + (c.numChans + 2);
   a.dct = f->d_dct; a.tw = f->d_tw; a.rtw = f->d_rtw; a.brev = f->d_brev; a.frameMean = f->d_frameMean; a.out = dOut;

   if (c.hasE || c.zMeanSource) {
      hipLaunchKernelGGL(k_mfcc_energy, dim3((F + 63) / 64), dim3(64), 0, s, a);
      HIPCHECK(hipGetLastError());
   }
   const size_t per = sizeof(float) * ((((size_t)t.fftN + 2 * ((size_t)t.fftN / 2 + 2) + c.numChans + 2) + 3) & ~(size_t)3);
   // (a grid of persistent wavefronts -- HTKAMD_MFCC_WPC per CU -- was tried against one workgroup per pair of frames: 1.93 ms at 16 or 32 per CU
   //  against 1.78 for the plain grid, tools/r06_mfcc2.sh; the loop stays, the default grid covers every pair)
   int wpc = 1 << 20;
   { const char *e = getenv("HTKAMD_MFCC_WPC"); if (e && atoi(e) > 0) wpc = atoi(e); }
   const int maxGrid = 256 * wpc;
   if (c.numChans <= 32 && c.numCeps <= 31 && !getenv("HTKAMD_MFCC_ONE_FRAME")) hipLaunchKernelGGL(k_mfcc_frames<true>, dim3(std::min((F + 1) / 2, maxGrid)), dim3(64), 2 * per, s, a);
   else hipLaunchKernelGGL(k_mfcc_frames<false>, dim3(std::min(F, maxGrid)), dim3(64), per, s, a);
   HIPCHECK(hipGetLastError());
   if (c.hasE && c.eNormalise) {
      hipLaunchKernelGGL(k_mfcc_enorm, dim3(nUtt), dim3(256), 0, s, dOut, f->d_frameOff, nCols, nStat - 1, c.silFloor, c.eScale);
      HIPCHECK(hipGetLastError());
   }
   if (c.hasD) {
      const size_t n = (size_t)F * nStat;
      hipLaunchKernelGGL(k_mfcc_delta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dOut, f->d_frameUtt, f->d_frameOff, F, nCols, 0, nStat, nStat, c.delWin);
      HIPCHECK(hipGetLastError());
      if (c.hasA) {
         hipLaunchKernelGGL(k_mfcc_delta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dOut, f->d_frameUtt, f->d_frameOff, F, nCols, nStat, 2 * nStat, nStat, c.accWin);
         HIPCHECK(hipGetLastError());
      }
   }
   if (c.hasZ) {
      const int d = c.numCeps + (c.hasC0 ? 1 : 0);
      hipLaunchKernelGGL(k_mfcc_zmean, dim3((nUtt * d + 63) / 64), dim3(64), 0, s, dOut, f->d_frameOff, nUtt, nCols, d);
      HIPCHECK(hipGetLastError());
   }
   HIPCHECK(hipStreamSynchronize(s));           // (sampOff / frameOff are the caller's: the copies above must be over when the call returns)
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ qualifiers on a parameterised table
__global__ void k_parm_widen(const float *in, float *out, size_t nFrames, int nStat, int nCols)
{
   const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= nFrames * (size_t)nStat) return;
   const size_t f = g / nStat;
   const int k = (int)(g % nStat);
   out[f * nCols + k] = in[g];
}

// _N: the row without its absolute energy / C0 column (ExtractObservation HParm.c:2882-2893)
__global__ void k_parm_drop_col(const float *in, float *out, size_t nFrames, int nFull, int col)
{
   const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   const int nCols = nFull - 1;
   if (g >= nFrames * (size_t)nCols) return;
   const size_t f = g / nCols;
   const int k = (int)(g % nCols);
   out[g] = in[f * nFull + (k < col ? k : k + 1)];
}

extern "C" int htkamd_parm_quals_cols(const htkamd_parm_quals *q)
{
   if (!q || q->nStat <= 0) return 0;
   return q->nStat * (1 + (q->hasD ? 1 : 0) + (q->hasA ? 1 : 0) + (q->hasT ? 1 : 0)) - (q->nullECol >= 0 ? 1 : 0);
}

extern "C" int htkamd_parm_qualify(const float *dStatic, const int *frameOff, int nUtt, const htkamd_parm_quals *q, float *dOut, void *stream)
{
   if (!q || !frameOff || nUtt < 0 || q->nStat <= 0 || (q->hasA && !q->hasD) || (q->hasT && !q->hasA) ||
       (q->hasD && q->delWin < 1) || (q->hasA && q->accWin < 1) || (q->hasT && q->thirdWin < 1) ||
       q->nZeroMean < 0 || q->nZeroMean > q->nStat || q->nullECol >= q->nStat || (q->nullECol >= 0 && !q->hasD)) {
      htkamd_set_error("parm_qualify: bad argument"); return HTKAMD_EINVAL;       /* _N needs _D: ValidConversion HParm.c:1420 */
   }
   const int F = nUtt ? frameOff[nUtt] : 0;
   if (F == 0) return HTKAMD_OK;
   if (!dStatic || !dOut) { htkamd_set_error("parm_qualify: NULL table"); return HTKAMD_EINVAL; }
   hipStream_t s = (hipStream_t)stream;
   const int nStat = q->nStat;
   const int nFull = nStat * (1 + (q->hasD ? 1 : 0) + (q->hasA ? 1 : 0) + (q->hasT ? 1 : 0));
   std::vector<int> frameUtt((size_t)F);
   for (int u = 0; u < nUtt; u++) {
      if (frameOff[u + 1] < frameOff[u]) { htkamd_set_error("parm_qualify: frameOff not monotone"); return HTKAMD_EINVAL; }
      for (int f = frameOff[u]; f < frameOff[u + 1]; f++) frameUtt[f] = u;
   }
   int *dUtt = nullptr, *dOff = nullptr;
   float *dFull = dOut;
   HIPCHECK(hipMalloc((void **)&dUtt, sizeof(int) * (size_t)F));
   HIPCHECK(hipMalloc((void **)&dOff, sizeof(int) * ((size_t)nUtt + 1)));
   if (q->nullECol >= 0) HIPCHECK(hipMalloc((void **)&dFull, sizeof(float) * (size_t)F * nFull));
   HIPCHECK(hipMemcpyAsync(dUtt, frameUtt.data(), sizeof(int) * (size_t)F, hipMemcpyHostToDevice, s));
   HIPCHECK(hipMemcpyAsync(dOff, frameOff, sizeof(int) * ((size_t)nUtt + 1), hipMemcpyHostToDevice, s));
   const size_t n = (size_t)F * nStat;
   const unsigned blocks = (unsigned)((n + 255) / 256);
   hipLaunchKernelGGL(k_parm_widen, dim3(blocks), dim3(256), 0, s, dStatic, dFull, (size_t)F, nStat, nFull);
   const int mode = (q->simpleDiffs ? 1 : 0) | (q->v1Compat ? 2 : 0);
   if (q->hasD) hipLaunchKernelGGL(k_mfcc_delta, dim3(blocks), dim3(256), 0, s, dFull, dUtt, dOff, F, nFull, 0, nStat, nStat, q->delWin, mode);
   if (q->hasA) hipLaunchKernelGGL(k_mfcc_delta, dim3(blocks), dim3(256), 0, s, dFull, dUtt, dOff, F, nFull, nStat, 2 * nStat, nStat, q->accWin, mode);
   if (q->hasT) hipLaunchKernelGGL(k_mfcc_delta, dim3(blocks), dim3(256), 0, s, dFull, dUtt, dOff, F, nFull, 2 * nStat, 3 * nStat, nStat, q->thirdWin, mode);
   if (q->nZeroMean > 0)
      hipLaunchKernelGGL(k_mfcc_zmean, dim3((nUtt * q->nZeroMean + 63) / 64), dim3(64), 0, s, dFull, dOff, nUtt, nFull, q->nZeroMean);
   if (q->nullECol >= 0) {
      const size_t m = (size_t)F * (nFull - 1);
      hipLaunchKernelGGL(k_parm_drop_col, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, dFull, dOut, (size_t)F, nFull, q->nullECol);
   }
   hipError_t e = hipGetLastError();
   hipError_t e2 = hipStreamSynchronize(s);
   (void)hipFree(dUtt); (void)hipFree(dOff);
   if (dFull != dOut) (void)hipFree(dFull);
   HIPCHECK(e); HIPCHECK(e2);
   return HTKAMD_OK;
}

extern "C" int htkamd_parm_add_qualifiers(const float *dStatic, const int *frameOff, int nUtt, int nStat, int hasD, int hasA,
                                          int delWin, int accWin, float *dOut, void *stream)
{
   if (nStat <= 0 || (hasA && !hasD) || delWin < 1 || accWin < 1) { htkamd_set_error("parm_add_qualifiers: bad argument"); return HTKAMD_EINVAL; }
   htkamd_parm_quals q;
   q.nStat = nStat; q.nZeroMean = 0; q.hasD = hasD; q.hasA = hasA; q.hasT = 0; q.delWin = delWin; q.accWin = accWin; q.thirdWin = 2; q.nullECol = -1; q.v1Compat = 0; q.simpleDiffs = 0;
   return htkamd_parm_qualify(dStatic, frameOff, nUtt, &q, dOut, stream);
}

// ------------------------------------------------------------------------------------ buffered qualifiers (HParm's buffer mode)
// FillBufFromChannel (HParm.c:4000-4116) qualifies the rows qst..qen of its buffer as soon as the qwin rows of look-ahead behind
// qen have arrived (tail = qwin, head = min(qst, qwin) rows of real left context; at the end of the input tail = 0), so that the
// regression windows of interior rows see real neighbours and only the first / last rows of the UTTERANCE replicate -- i.e. exactly
// the table-mode values, delivered qwin rows late.  Same here: every push qualifies the window [qst - head, rows seen) as one
// utterance on the device and hands out its rows qst..qen.
struct htkamd_parm_stream {
   htkamd_parm_quals q;
   int qwin, nSeen, qst, base, cap;        // rows seen so far; next row to hand out; absolute index of dBuf's first row; rows of capacity
   int done;
   float *dBuf, *dTmp;
};

extern "C" int htkamd_parm_stream_open(const htkamd_parm_quals *q, int maxRows, htkamd_parm_stream **out)
{
   if (!q || !out || maxRows < 1 || q->nStat <= 0) { htkamd_set_error("parm_stream_open: bad argument"); return HTKAMD_EINVAL; }
   if (q->nZeroMean > 0) { htkamd_set_error("parm_stream_open: _Z needs the whole utterance (table mode: htkamd_parm_qualify)"); return HTKAMD_EINVAL; }
   if ((q->hasA && !q->hasD) || (q->hasT && !q->hasA) || (q->hasD && q->delWin < 1) || (q->hasA && q->accWin < 1) || (q->hasT && q->thirdWin < 1) ||
       q->nullECol >= q->nStat || (q->nullECol >= 0 && !q->hasD)) { htkamd_set_error("parm_stream_open: bad qualifier set"); return HTKAMD_EINVAL; }
   int nd = 0;
   if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) { htkamd_set_error("parm_stream_open: no HIP device"); return HTKAMD_ENODEV; }
   htkamd_parm_stream *s = new htkamd_parm_stream();
   s->q = *q;
   s->qwin = (q->hasD ? q->delWin : 0) + (q->hasA ? q->accWin : 0) + (q->hasT ? q->thirdWin : 0);
   s->nSeen = s->qst = s->base = 0; s->done = 0;
   s->cap = maxRows + 2 * s->qwin + 1;
   s->dBuf = s->dTmp = nullptr;
   const int cols = htkamd_parm_quals_cols(q);
   if (hipMalloc((void **)&s->dBuf, sizeof(float) * (size_t)s->cap * q->nStat) != hipSuccess ||
       hipMalloc((void **)&s->dTmp, sizeof(float) * (size_t)s->cap * cols) != hipSuccess) {
      if (s->dBuf) (void)hipFree(s->dBuf);
      delete s;
      htkamd_set_error("parm_stream_open: out of device memory"); return HTKAMD_ENOMEM;
   }
   *out = s;
   return HTKAMD_OK;
}

extern "C" void htkamd_parm_stream_close(htkamd_parm_stream *s)
{
   if (!s) return;
   if (s->dBuf) (void)hipFree(s->dBuf);
   if (s->dTmp) (void)hipFree(s->dTmp);
   delete s;
}

extern "C" int htkamd_parm_stream_lookahead(const htkamd_parm_stream *s) { return s ? s->qwin : 0; }

extern "C" int htkamd_parm_stream_push(htkamd_parm_stream *s, const float *dStatic, int nRows, int last, float *dOut, int *nOut, void *stream)
{
   if (!s || nRows < 0 || !nOut || (nRows > 0 && !dStatic)) { htkamd_set_error("parm_stream_push: bad argument"); return HTKAMD_EINVAL; }
   *nOut = 0;
   if (s->done) { htkamd_set_error("parm_stream_push: the stream was closed by a push with last = 1"); return HTKAMD_EINVAL; }
   hipStream_t st = (hipStream_t)stream;
   const int nStat = s->q.nStat, cols = htkamd_parm_quals_cols(&s->q);
   if (s->nSeen - s->base + nRows > s->cap) { htkamd_set_error("parm_stream_push: %d rows exceed the stream's maxRows", nRows); return HTKAMD_EINVAL; }
   if (nRows > 0) HIPCHECK(hipMemcpyAsync(s->dBuf + (size_t)(s->nSeen - s->base) * nStat, dStatic, sizeof(float) * (size_t)nRows * nStat, hipMemcpyDeviceToDevice, st));
   s->nSeen += nRows;
   if (last) s->done = 1;
   const int qen = last ? s->nSeen - 1 : s->nSeen - s->qwin - 1;
   if (qen < s->qst) return HTKAMD_OK;
   if (!dOut) { htkamd_set_error("parm_stream_push: NULL output"); return HTKAMD_EINVAL; }
   const int head = (s->qst < s->qwin) ? s->qst : s->qwin;
   const int w0 = s->qst - head;
   const int frameOff[2] = {0, s->nSeen - w0};
   int rc = htkamd_parm_qualify(s->dBuf + (size_t)(w0 - s->base) * nStat, frameOff, 1, &s->q, s->dTmp, stream);
   if (rc) return rc;
   const int n = qen - s->qst + 1;
   HIPCHECK(hipMemcpyAsync(dOut, s->dTmp + (size_t)head * cols, sizeof(float) * (size_t)n * cols, hipMemcpyDeviceToDevice, st));
   *nOut = n;
   s->qst = qen + 1;
   // keep qwin rows of left context for the next window
   const int keep0 = (s->qst - s->qwin > s->base) ? s->qst - s->qwin : s->base;
   if (keep0 > s->base) {
      const int nKeep = s->nSeen - keep0;
      if (nKeep > 0) {           // overlapping ranges: through the scratch table
         HIPCHECK(hipMemcpyAsync(s->dTmp, s->dBuf + (size_t)(keep0 - s->base) * nStat, sizeof(float) * (size_t)nKeep * nStat, hipMemcpyDeviceToDevice, st));
         HIPCHECK(hipStreamSynchronize(st));      // dOut was copied out of dTmp above
         HIPCHECK(hipMemcpyAsync(s->dBuf, s->dTmp, sizeof(float) * (size_t)nKeep * nStat, hipMemcpyDeviceToDevice, st));
      }
      s->base = keep0;
   }
   HIPCHECK(hipStreamSynchronize(st));
   return HTKAMD_OK;
}

// ------------------------------------------------------------------------------------ HCompV: global mean and variance
// One block per slab of frames, thread = dimension (strided), fp64 partial sums, one atomic per (block, dimension).
__global__ void k_compv(const float *X, long long nFrames, int D, double *acc)
{
   const long long per = (nFrames + gridDim.x - 1) / gridDim.x;
   const long long f0 = (long long)blockIdx.x * per, f1 = (f0 + per < nFrames) ? f0 + per : nFrames;
   for (int k = threadIdx.x; k < D; k += blockDim.x) {
      double s = 0.0, q = 0.0;
      for (long long f = f0; f < f1; f++) { const double v = (double)X[(size_t)f * D + k]; s += v; q += v * v; }
      if (f1 > f0) { atomicAdd(acc + k, s); atomicAdd(acc + D + k, q); }
   }
}

extern "C" int htkamd_compv(const float *dX, long long nFrames, int D, float minVar, float *mean, float *var, void *stream)
{
   if (!dX || !mean || !var || D <= 0) { htkamd_set_error("compv: bad argument"); return HTKAMD_EINVAL; }
   if (nFrames < 2) { htkamd_set_error("compv: only %lld frames (HCompV error 2021)", nFrames); return HTKAMD_EINVAL; }
   hipStream_t s = (hipStream_t)stream;
   double *dAcc = nullptr;
   HIPCHECK(hipMalloc((void **)&dAcc, sizeof(double) * 2 * (size_t)D));
   HIPCHECK(hipMemsetAsync(dAcc, 0, sizeof(double) * 2 * (size_t)D, s));
   long long blocks = (nFrames + 255) / 256;
   if (blocks > 4096) blocks = 4096;
   hipLaunchKernelGGL(k_compv, dim3((unsigned)blocks), dim3(64), 0, s, dX, nFrames, D, dAcc);
   std::vector<double> h(2 * (size_t)D);
   hipError_t e = hipGetLastError();
   hipError_t e2 = hipMemcpyAsync(h.data(), dAcc, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s);
   hipError_t e3 = hipStreamSynchronize(s);
   (void)hipFree(dAcc);
   HIPCHECK(e); HIPCHECK(e2); HIPCHECK(e3);
   const double n = (double)nFrames;
   for (int k = 0; k < D; k++) {
      const double m = h[k] / n, v = h[D + k] / n - m * m;
      mean[k] = (float)m;
      var[k] = ((float)v > minVar) ? (float)v : minVar;
   }
   return HTKAMD_OK;
}
