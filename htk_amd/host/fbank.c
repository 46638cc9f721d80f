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

static float mel_of(int k, float fres) { return 1127 * log(1 + (k - 1) * fres); }

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
   int fftN = 2, Nby2, maxChan, k, chan, i, j, b;
   float fres, mlo, mhi, ms, *cf;
   short *loChan;
   memset(t, 0, sizeof(*t));
   t->frSize = (int)(c->winDur / c->sampPeriod);
   t->frRate = (int)(c->frPeriod / c->sampPeriod);
   if (t->frSize < 2 || t->frRate < 1 || c->numChans < 1 || c->numCeps < 1 || c->numCeps > 64 || c->numChans > 63) {
      htkamd_set_error("mfcc: unsupported geometry (frSize %d frRate %d chans %d ceps %d)", t->frSize, t->frRate, c->numChans, c->numCeps);
      return HTKAMD_EINVAL;
   }
   while (t->frSize > fftN) fftN *= 2;
   if (fftN < 8 || fftN > 4096) { htkamd_set_error("mfcc: FFT size %d outside 8..4096", fftN); return HTKAMD_EINVAL; }
   t->fftN = fftN; Nby2 = fftN / 2; maxChan = c->numChans + 1;
   fres = 1.0E7 / ((long)c->sampPeriod * fftN * 700.0);
   t->klo = 2; t->khi = Nby2;
   mlo = 0; mhi = mel_of(Nby2 + 1, fres);
   if (c->loFreq >= 0.0) {
      mlo = 1127 * log(1 + c->loFreq / 700.0);
      t->klo = (int)((c->loFreq * (long)c->sampPeriod * 1.0e-7 * fftN) + 2.5);
      if (t->klo < 2) t->klo = 2;
   }
   if (c->hiFreq >= 0.0) {
      mhi = 1127 * log(1 + c->hiFreq / 700.0);
      t->khi = (int)((c->hiFreq * (long)c->sampPeriod * 1.0e-7 * fftN) + 0.5);
      if (t->khi > Nby2) t->khi = Nby2;
   }
   cf = (float *)malloc(sizeof(float) * (size_t)(maxChan + 2));
   ms = mhi - mlo;
   for (chan = 1; chan <= maxChan; chan++) cf[chan] = ((float)chan / (float)maxChan) * ms + mlo;
   loChan = (short *)malloc(sizeof(short) * (size_t)(Nby2 + 2));
   for (k = 1, chan = 1; k <= Nby2; k++) {
      const float melk = mel_of(k, fres);
      if (k < t->klo || k > t->khi) loChan[k] = -1;
      else {
         while (cf[chan] < melk && chan <= maxChan) ++chan;
         loChan[k] = (short)(chan - 1);
      }
   }
   t->loWt = (float *)calloc((size_t)Nby2 + 2, sizeof(float));
   for (k = 1; k <= Nby2; k++) {
      chan = loChan[k];
      if (k < t->klo || k > t->khi) t->loWt[k] = 0.0;
      else if (chan > 0) t->loWt[k] = ((cf[chan + 1] - mel_of(k, fres)) / (cf[chan + 1] - cf[chan]));
      else t->loWt[k] = (cf[1] - mel_of(k, fres)) / (cf[1] - mlo);
   }
   /* Wave2FBank adds, for k = klo..khi in order, loWt*ek to bin loChan[k] and ek-loWt*ek to bin loChan[k]+1.
      loChan is non-decreasing in k, so bin b receives first the (ek - t1) terms of the k with loChan == b-1 and
      then the t1 terms of the k with loChan == b: two contiguous k ranges per bin. */
   t->binA0 = (int *)calloc((size_t)4 * (c->numChans + 2), sizeof(int));
   t->binA1 = t->binA0 + (c->numChans + 2); t->binB0 = t->binA1 + (c->numChans + 2); t->binB1 = t->binB0 + (c->numChans + 2);
   for (b = 1; b <= c->numChans; b++) { t->binA0[b] = 1; t->binA1[b] = 0; t->binB0[b] = 1; t->binB1[b] = 0; }
   for (k = t->klo; k <= t->khi; k++) {
      const int bin = loChan[k];
      if (bin > 0) { if (t->binB1[bin] < t->binB0[bin]) t->binB0[bin] = k; t->binB1[bin] = k; }
      if (bin < c->numChans) { if (t->binA1[bin + 1] < t->binA0[bin + 1]) t->binA0[bin + 1] = k; t->binA1[bin + 1] = k; }
   }
   free(cf); free(loChan);
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
   /* complex FFT of nn = fftN/2 points: stage with half-size h uses twiddles (wr, wi)[0..h-1] */
   {
      const int nn = fftN / 2;
      int limit, off = 0, bits = 0;
      t->tw = (double *)calloc((size_t)2 * nn, sizeof(double));
      for (limit = 2; limit < fftN; limit *= 2) {
         const double theta = HTK_TPI / limit, x = sin(0.5 * theta);
         const double wpr = -2.0 * x * x, wpi = sin(theta);
         double wr = 1.0, wi = 0.0, wx;
         int ii;
         for (ii = 1; ii <= limit / 2; ii++) {
            t->tw[2 * (off + ii - 1)] = wr; t->tw[2 * (off + ii - 1) + 1] = wi;
            wx = wr;
            wr = wr * wpr - wi * wpi + wr;
            wi = wi * wpr + wx * wpi + wi;
         }
         off += limit / 2;
      }
      /* Realft post-pass: (yr, yi) for i = 2..n2 */
      {
         const int n = fftN / 2, n2 = n / 2;
         const double theta = HTK_PI / n, x = sin(0.5 * theta);
         const double yr2 = -2.0 * x * x, yi2 = sin(theta);
         double yr = 1.0 + yr2, yi = yi2, yr0;
         t->rtw = (double *)calloc((size_t)2 * (n2 + 2), sizeof(double));
         for (i = 2; i <= n2; i++) {
            t->rtw[2 * i] = yr; t->rtw[2 * i + 1] = yi;
            yr0 = yr;
            yr = yr * yr2 - yi * yi2 + yr;
            yi = yi * yr2 + yr0 * yi2 + yi;
         }
      }
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
