/* fbank.c -- host-side (C) tables of the MFCC front end: everything the reference computes once per
 * configuration with libm is computed here the same way and handed to the device as tables, so that the
 * per-frame kernels only do the reference's multiply/add sequences:
 *   frame geometry        HWave.c:1575-1576 (frSize, frRate), :1663 FramesInWave
 *   mel filterbank        InitFBank HSigP.c:471-555 (fres, centre frequencies, loChan, loWt) -> per-bin k ranges
 *   Hamming window        GenHamWindow HSigP.c:108-120
 *   lifter                GenCepWin HSigP.c:755-770
 *   DCT cosines           FBank2MFCC HSigP.c:607-621 (cos(x*(k-0.5)) per term, double)
 *   FFT twiddles          the double-precision recurrences of FFT HSigP.c:332-349 and Realft :371-386, tabulated
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

#define HTK_PI 3.14159265358979

/* mel value of FFT bin k (1-based): Mel(k, fres) HSigP.c:443 */
static float bin_mel(int k, float fres) { return 1127 * log(1 + (k - 1) * fres); }
static float hz_mel(float hz) { return 1127 * log(1 + hz / 700.0); }

/* The reference generates its FFT and Realft twiddles by walking round the unit circle in double:
      w_0 = 1,  w_{n+1} = w_n + w_n * ((cos t - 1) + i sin t),   cos t - 1 = -2 sin^2(t/2)
   (HSigP.c:332-349, :371-386).  Emits steps first .. first+count-1 of that walk as (re, im) pairs, with the same double operations
   in the same order, so that the tabulated values are the ones the reference multiplies with. */
static void unit_walk(double theta, int first, int count, double *out)
{
   const double h = sin(0.5 * theta), dRe = -2.0 * h * h, dIm = sin(theta);
   double re = 1.0, im = 0.0;
   int n;
   for (n = 0; n < first + count; n++) {
      const double re0 = re;
      if (n >= first) { out[2 * (n - first)] = re; out[2 * (n - first) + 1] = im; }
      re = re * dRe - im * dIm + re;
      im = im * dRe + re0 * dIm + im;
   }
}

int htkamd_mfcc_num_frames(const htkamd_mfcc_config *c, int nSamples)
{
   const int fs = (int)(c->winDur / c->sampPeriod), fr = (int)(c->frPeriod / c->sampPeriod);
   if (fs <= 0 || fr <= 0 || fs > nSamples) return 0;
   return (nSamples - fs) / fr + 1;
}

int htkamd_mfcc_num_cols(const htkamd_mfcc_config *c)
{
   const int nStat = c->numCeps + (c->hasC0 ? 1 : 0) + (c->hasE ? 1 : 0);
   return nStat * (1 + (c->hasD ? 1 : 0) + (c->hasA ? 1 : 0));
}

void htkamd_mfcc_tables_free(struct htkamd_mfcc_tables *t)
{
   free(t->ham); free(t->cepWin); free(t->loWt); free(t->binA0); free(t->dct); free(t->tw); free(t->rtw); free(t->brev);
   memset(t, 0, sizeof(*t));
}

int htkamd_mfcc_tables_build(const htkamd_mfcc_config *c, struct htkamd_mfcc_tables *t)
{
   int fftN = 2, half, nEdge, k, i, j, b;
   float fres, melLo, melHi, *edge;
   memset(t, 0, sizeof(*t));
   t->frSize = (int)(c->winDur / c->sampPeriod);
   t->frRate = (int)(c->frPeriod / c->sampPeriod);
   if (t->frSize < 2 || t->frRate < 1 || c->numChans < 1 || c->numCeps < 1 || c->numCeps > 64 || c->numChans > 63) {
      htkamd_set_error("mfcc: unsupported geometry (frSize %d frRate %d chans %d ceps %d)", t->frSize, t->frRate, c->numChans, c->numCeps);
      return HTKAMD_EINVAL;
   }
   while (t->frSize > fftN) fftN *= 2;
   if (fftN < 8 || fftN > 4096) { htkamd_set_error("mfcc: FFT size %d outside 8..4096", fftN); return HTKAMD_EINVAL; }
   t->fftN = fftN; half = fftN / 2;
   /* ---- mel filterbank (InitFBank HSigP.c:471-555 supplies the numbers; the tables are this file's own form).
      Band of interest: FFT bins klo..khi, mel range melLo..melHi, optionally narrowed by LOFREQ / HIFREQ. */
   fres = 1.0E7 / ((long)c->sampPeriod * fftN * 700.0);
   t->klo = 2; t->khi = half;
   melLo = 0; melHi = bin_mel(half + 1, fres);
   if (c->loFreq >= 0.0) {
      melLo = hz_mel(c->loFreq);
      t->klo = (int)((c->loFreq * (long)c->sampPeriod * 1.0e-7 * fftN) + 2.5);
      if (t->klo < 2) t->klo = 2;
   }
   if (c->hiFreq >= 0.0) {
      melHi = hz_mel(c->hiFreq);
      t->khi = (int)((c->hiFreq * (long)c->sampPeriod * 1.0e-7 * fftN) + 0.5);
      if (t->khi > half) t->khi = half;
   }
   /* numChans triangular filters share numChans+2 equally spaced mel edges: edge[0] = melLo, edge[e] = e/(numChans+1) of the span */
   nEdge = c->numChans + 1;
   edge = (float *)malloc(sizeof(float) * (size_t)(nEdge + 2));
   edge[0] = melLo;
   for (b = 1; b <= nEdge; b++) edge[b] = ((float)b / (float)nEdge) * (melHi - melLo) + melLo;
   edge[nEdge + 1] = edge[nEdge] + 1.0f;                /* guard: a top bin whose centre rounds above HIFREQ feeds no filter */
   /* An FFT bin k between edge[lo] and edge[lo+1] feeds filter lo with weight loWt[k] = (edge[lo+1] - mel)/(edge[lo+1] - edge[lo]) and
      filter lo+1 with the rest (Wave2FBank HSigP.c:558-604 adds loWt*ek to bin lo and ek - loWt*ek to bin lo+1, for k ascending).
      lo never decreases with k, so filter b receives, in this order, the "rest" terms of the bins with lo == b-1 and then the
      weighted terms of the bins with lo == b: two contiguous k ranges per filter, [binA0,binA1] and [binB0,binB1]. */
   t->loWt = (float *)calloc((size_t)half + 2, sizeof(float));
   t->binA0 = (int *)calloc((size_t)4 * (c->numChans + 2), sizeof(int));
   t->binA1 = t->binA0 + (c->numChans + 2); t->binB0 = t->binA1 + (c->numChans + 2); t->binB1 = t->binB0 + (c->numChans + 2);
   for (b = 1; b <= c->numChans; b++) { t->binA0[b] = 1; t->binA1[b] = 0; t->binB0[b] = 1; t->binB1[b] = 0; }
   {
      int lo = 0;                                        /* edges 1..lo lie below the current bin's mel value */
      for (k = 1; k <= half; k++) {
         const float mel = bin_mel(k, fres);
         if (k < t->klo || k > t->khi) continue;        /* outside the band: weight 0, feeds nothing */
         while (lo < nEdge && edge[lo + 1] < mel) lo++;
         t->loWt[k] = (edge[lo + 1] - mel) / (edge[lo + 1] - edge[lo]);
         if (lo >= 1 && lo <= c->numChans) { if (t->binB1[lo] < t->binB0[lo]) t->binB0[lo] = k; t->binB1[lo] = k; }
         if (lo < c->numChans) { if (t->binA1[lo + 1] < t->binA0[lo + 1]) t->binA0[lo + 1] = k; t->binA1[lo + 1] = k; }
      }
   }
   free(edge);
   t->ham = (float *)calloc((size_t)t->frSize + 1, sizeof(float));
   { const float a = HTK_TPI / (t->frSize - 1); for (i = 1; i <= t->frSize; i++) t->ham[i] = 0.54 - 0.46 * cos(a * (i - 1)); }
   t->cepWin = (float *)calloc((size_t)c->numCeps + 1, sizeof(float));
   for (i = 1; i <= c->numCeps; i++) t->cepWin[i] = 1.0f;
   if (c->cepLifter > 0) {
      const float a = HTK_PI / c->cepLifter, Lby2 = c->cepLifter / 2.0;
      for (i = 1; i <= c->numCeps; i++) t->cepWin[i] = 1.0 + Lby2 * sin(i * a);
   }
   t->mfnorm = sqrt(2.0 / (float)c->numChans);
   t->dct = (double *)calloc((size_t)(c->numCeps + 1) * (c->numChans + 1), sizeof(double));
   {
      const float pi_factor = HTK_PI / (float)c->numChans;
      for (j = 1; j <= c->numCeps; j++) {
         const float x = (float)j * pi_factor;
         for (k = 1; k <= c->numChans; k++) t->dct[(size_t)j * (c->numChans + 1) + k] = cos(x * (k - 0.5));
      }
   }
   /* complex FFT of nn = fftN/2 points: the stage that combines blocks of `limit/2` points uses steps 0..limit/2-1 of the walk with
      angle 2 pi/limit; Realft's post-pass (i = 2..nn/2) steps 1.. of the walk with angle pi/nn */
   {
      const int nn = fftN / 2;
      int limit, off = 0, bits = 0;
      t->tw = (double *)calloc((size_t)2 * nn, sizeof(double));
      for (limit = 2; limit < fftN; limit *= 2) {
         unit_walk(HTK_TPI / limit, 0, limit / 2, t->tw + 2 * off);
         off += limit / 2;
      }
      t->rtw = (double *)calloc((size_t)2 * (nn / 2 + 2), sizeof(double));
      if (nn / 2 >= 2) unit_walk(HTK_PI / nn, 1, nn / 2 - 1, t->rtw + 4);
      /* bit reversal of the complex index (the swap loop of HSigP.c:319-331) */
      while ((1 << bits) < nn) bits++;
      t->brev = (short *)calloc((size_t)nn, sizeof(short));
      for (i = 0; i < nn; i++) {
         int r = 0;
         for (j = 0; j < bits; j++) if (i & (1 << j)) r |= 1 << (bits - 1 - j);
         t->brev[i] = (short)r;
      }
   }
   return HTKAMD_OK;
}
