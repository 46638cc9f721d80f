/* orc_mfcc.c -- CPU restatement of the reference's waveform -> MFCC(+C0/E)(+D)(+A)(+Z) front end
 * (TEST INFRASTRUCTURE): HWave frame slicing, HParm ConvertFrame, HSigP primitives, AddQualifiers.
 * Types and operation order follow the reference (float data, double twiddle recurrences and libm calls),
 * so with the same libm the output equals HCopy's bit for bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "htk_oracle.h"

/* HSigP.c:311-358 FFT (forward only), 1-based array s[1..n] of (re,im) pairs */
static void fft_fwd(float *s, int n)
{
   int ii, jj, nn, limit, m, j, inc, i;
   double wx, wr, wpr, wpi, wi, theta;
   double xre, xri, x;
   nn = n / 2; j = 1;
   for (ii = 1; ii <= nn; ii++) {
      i = 2 * ii - 1;
      if (j > i) {
         xre = s[j]; xri = s[j + 1];
         s[j] = s[i]; s[j + 1] = s[i + 1];
         s[i] = xre; s[i + 1] = xri;
      }
      m = n / 2;
      while (m >= 2 && j > m) { j -= m; m /= 2; }
      j += m;
   }
   limit = 2;
   while (limit < n) {
      inc = 2 * limit; theta = ORC_TPI / limit;
      x = sin(0.5 * theta);
      wpr = -2.0 * x * x; wpi = sin(theta);
      wr = 1.0; wi = 0.0;
      for (ii = 1; ii <= limit / 2; ii++) {
         m = 2 * ii - 1;
         for (jj = 0; jj <= (n - m) / inc; jj++) {
            i = m + jj * inc;
            j = i + limit;
            xre = wr * s[j] - wi * s[j + 1];
            xri = wr * s[j + 1] + wi * s[j];
            s[j] = s[i] - xre; s[j + 1] = s[i + 1] - xri;
            s[i] = s[i] + xre; s[i + 1] = s[i + 1] + xri;
         }
         wx = wr;
         wr = wr * wpr - wi * wpi + wr;
         wi = wi * wpr + wx * wpi + wi;
      }
      limit = inc;
   }
}

/* HSigP.c:362-391 Realft, s[1..2n] */
static void realft(float *s, int size)
{
   int n, n2, i, i1, i2, i3, i4;
   double xr1, xi1, xr2, xi2, wrs, wis;
   double yr, yi, yr2, yi2, yr0, theta, x;
   n = size / 2; n2 = n / 2;
   theta = ORC_PI / n;
   fft_fwd(s, size);
   x = sin(0.5 * theta);
   yr2 = -2.0 * x * x;
   yi2 = sin(theta); yr = 1.0 + yr2; yi = yi2;
   for (i = 2; i <= n2; i++) {
      i1 = i + i - 1; i2 = i1 + 1;
      i3 = n + n + 3 - i2; i4 = i3 + 1;
      wrs = yr; wis = yi;
      xr1 = (s[i1] + s[i3]) / 2.0; xi1 = (s[i2] - s[i4]) / 2.0;
      xr2 = (s[i2] + s[i4]) / 2.0; xi2 = (s[i3] - s[i1]) / 2.0;
      s[i1] = xr1 + wrs * xr2 - wis * xi2;
      s[i2] = xi1 + wrs * xi2 + wis * xr2;
      s[i3] = xr1 - wrs * xr2 + wis * xi2;
      s[i4] = -xi1 + wrs * xi2 + wis * xr2;
      yr0 = yr;
      yr = yr * yr2 - yi * yi2 + yr;
      yi = yi * yr2 + yr0 * yi2 + yi;
   }
   xr1 = s[1];
   s[1] = xr1 + s[2];
   s[2] = 0.0;
}

static float mel(int k, float fres) { return 1127 * log(1 + (k - 1) * fres); }   /* HSigP.c:443 */

/* HSigP.c:827-856 Regress */
static void regress(float *data, int vSize, int n, int step, int offset, int delwin, int head, int tail)
{
   float *fp, *fp1, *fp2, *back, *forw;
   float sum, sigmaT2;
   int i, t, j;
   sigmaT2 = 0.0;
   for (t = 1; t <= delwin; t++) sigmaT2 += t * t;
   sigmaT2 *= 2.0;
   fp = data;
   for (i = 1; i <= n; i++) {
      fp1 = fp; fp2 = fp + offset;
      for (j = 1; j <= vSize; j++) {
         back = forw = fp1; sum = 0.0;
         for (t = 1; t <= delwin; t++) {
            if (head + i - t > 0) back -= step;
            if (tail + n - i + 1 - t > 0) forw += step;
            sum += t * (*forw - *back);
         }
         *fp2 = sum / sigmaT2;
         ++fp1; ++fp2;
      }
      fp += step;
   }
}

/* HParm.c:1552-1598 AddDiffs for a whole table (hdMargin = tlMargin = 0, not V1COMPAT, regression form) */
static void add_diffs(float *data, int nRows, int nCols, int si, int ti, int d, int winSize)
{
   float *p;
   int n, offset = ti - si, head = winSize, tail = winSize;
   p = data + si; n = nRows - (head + tail);
   if (n <= 0) {                                      /* ultra short: one call with both margins (HParm.c:1568-1573) */
      regress(p, d, nRows, nCols, offset, winSize, 0, 0);
      return;
   }
   regress(p, d, head, nCols, offset, winSize, 0, winSize);
   p += head * nCols;
   regress(p, d, n, nCols, offset, winSize, winSize, winSize);
   p += n * nCols;
   regress(p, d, tail, nCols, offset, winSize, winSize, 0);
}

/* number of frames: HWave.c:1575-1576,1663-1668 */
int orc_mfcc_frames(int nSamples, const orc_mfcc_cfg *c, int *frSize, int *frRate)
{
   int fs = (int)(c->winDur / c->sampPeriod), fr = (int)(c->frPeriod / c->sampPeriod);
   if (frSize) *frSize = fs;
   if (frRate) *frRate = fr;
   if (fs > nSamples) return 0;
   return (nSamples - fs) / fr + 1;
}

int orc_mfcc_cols(const orc_mfcc_cfg *c)
{
   int nStat = c->numCeps + (c->hasC0 ? 1 : 0) + (c->hasE ? 1 : 0);
   return nStat * (1 + (c->hasD ? 1 : 0) + (c->hasA ? 1 : 0));
}

/* Whole-file conversion (OpenBuffer with maxObs == 0): out[T][cols] row-major.  Returns T. */
int orc_mfcc(const short *wav, int nSamples, const orc_mfcc_cfg *c, float *out)
{
   int frSize, frRate, T = orc_mfcc_frames(nSamples, c, &frSize, &frRate);
   int fftN = 2, Nby2, numChans = c->numChans, maxChan = numChans + 1, klo, khi, k, chan, i, j, t;
   int nStat = c->numCeps + (c->hasC0 ? 1 : 0) + (c->hasE ? 1 : 0), nCols = orc_mfcc_cols(c);
   float fres, mlo, mhi, ms, melk, *cf, *loWt, *s, *x, *fbank, *ham, *cepWin, *cc;
   short *loChan;
   if (T <= 0) return 0;
   /* InitFBank HSigP.c:471-555 */
   while (frSize > fftN) fftN *= 2;
   Nby2 = fftN / 2;
   fres = 1.0E7 / (c->sampPeriod * fftN * 700.0);
   klo = 2; khi = Nby2;
   mlo = 0; mhi = mel(Nby2 + 1, fres);
   if (c->loFreq >= 0.0) {
      mlo = 1127 * log(1 + c->loFreq / 700.0);
      klo = (int)((c->loFreq * c->sampPeriod * 1.0e-7 * fftN) + 2.5);
      if (klo < 2) klo = 2;
   }
   if (c->hiFreq >= 0.0) {
      mhi = 1127 * log(1 + c->hiFreq / 700.0);
      khi = (int)((c->hiFreq * c->sampPeriod * 1.0e-7 * fftN) + 0.5);
      if (khi > Nby2) khi = Nby2;
   }
   cf = (float *)malloc(sizeof(float) * (maxChan + 2));
   ms = mhi - mlo;
   for (chan = 1; chan <= maxChan; chan++) cf[chan] = ((float)chan / (float)maxChan) * ms + mlo;
   loChan = (short *)malloc(sizeof(short) * (Nby2 + 2));
   for (k = 1, chan = 1; k <= Nby2; k++) {
      melk = mel(k, fres);
      if (k < klo || k > khi) loChan[k] = -1;
      else {
         while (cf[chan] < melk && chan <= maxChan) ++chan;
         loChan[k] = chan - 1;
      }
   }
   loWt = (float *)malloc(sizeof(float) * (Nby2 + 2));
   for (k = 1; k <= Nby2; k++) {
      chan = loChan[k];
      if (k < klo || k > khi) loWt[k] = 0.0;
      else {
         if (chan > 0) loWt[k] = ((cf[chan + 1] - mel(k, fres)) / (cf[chan + 1] - cf[chan]));
         else loWt[k] = (cf[1] - mel(k, fres)) / (cf[1] - mlo);
      }
   }
   /* GenHamWindow HSigP.c:108-120 */
   ham = (float *)malloc(sizeof(float) * (frSize + 1));
   { float a = ORC_TPI / (frSize - 1); for (i = 1; i <= frSize; i++) ham[i] = 0.54 - 0.46 * cos(a * (i - 1)); }
   /* GenCepWin HSigP.c:755-770 */
   cepWin = (float *)malloc(sizeof(float) * (c->numCeps + 1));
   if (c->cepLifter > 0) {
      float a = ORC_PI / c->cepLifter, Lby2 = c->cepLifter / 2.0;
      for (i = 1; i <= c->numCeps; i++) cepWin[i] = 1.0 + Lby2 * sin(i * a);
   }
   s = (float *)malloc(sizeof(float) * (frSize + 1));
   x = (float *)malloc(sizeof(float) * (fftN + 1));
   fbank = (float *)malloc(sizeof(float) * (numChans + 1));
   cc = (float *)malloc(sizeof(float) * (c->numCeps + 1));

   for (t = 0; t < T; t++) {
      float *p = out + (size_t)t * nCols, rawte = 0.0, te = 0.0;
      for (k = 0; k < frSize; k++) s[k + 1] = wav[(size_t)t * frRate + k];          /* GetWave HWave.c:1683 */
      /* ConvertFrame HParm.c:2214-2305 */
      if (c->zMeanSource) {                                                          /* ZeroMeanFrame HParm.c:2132 */
         float sum = 0.0, off;
         for (i = 1; i <= frSize; i++) sum += s[i];
         off = sum / frSize;
         for (i = 1; i <= frSize; i++) s[i] -= off;
      }
      if (c->hasE && c->rawEnergy) {
         rawte = 0.0;
         for (i = 1; i <= frSize; i++) rawte += s[i] * s[i];
      }
      if (c->preEmph > 0.0) {                                                        /* PreEmphasise HSigP.c:134 */
         float preE = c->preEmph;
         for (i = frSize; i >= 2; i--) s[i] -= s[i - 1] * preE;
         s[1] *= 1.0 - preE;
      }
      if (c->useHam) for (i = 1; i <= frSize; i++) s[i] *= ham[i];
      /* Wave2FBank HSigP.c:558-604 */
      if (!(c->hasE && c->rawEnergy)) { te = 0.0; for (k = 1; k <= frSize; k++) te += (s[k] * s[k]); }
      for (k = 1; k <= frSize; k++) x[k] = s[k];
      for (k = frSize + 1; k <= fftN; k++) x[k] = 0.0;
      realft(x, fftN);
      for (i = 1; i <= numChans; i++) fbank[i] = 0.0;
      for (k = klo; k <= khi; k++) {
         float t1 = x[2 * k - 1], t2 = x[2 * k], ek;
         int bin;
         if (c->usePower) ek = t1 * t1 + t2 * t2;
         else ek = sqrt(t1 * t1 + t2 * t2);
         bin = loChan[k];
         t1 = loWt[k] * ek;
         if (bin > 0) fbank[bin] += t1;
         if (bin < numChans) fbank[bin + 1] += ek - t1;
      }
      for (i = 1; i <= numChans; i++) {
         float t1 = fbank[i];
         if (t1 < 1.0) t1 = 1.0;
         fbank[i] = log(t1);
      }
      /* FBank2MFCC HSigP.c:607-621 */
      {
         float mfnorm = sqrt(2.0 / (float)numChans), pi_factor = ORC_PI / (float)numChans, xx;
         for (j = 1; j <= c->numCeps; j++) {
            cc[j] = 0.0; xx = (float)j * pi_factor;
            for (k = 1; k <= numChans; k++) cc[j] += fbank[k] * cos(xx * (k - 0.5));
            cc[j] *= mfnorm;
         }
      }
      if (c->cepLifter > 0) for (i = 1; i <= c->numCeps; i++) cc[i] *= cepWin[i];    /* WeightCepstrum HSigP.c:773 */
      for (i = 1; i <= c->numCeps; i++) *p++ = cc[i] * c->cepScale;
      if (c->hasC0) {                                                                 /* FBank2C0 HSigP.c:647 */
         float mfnorm = sqrt(2.0 / (float)numChans), sum = 0.0;
         for (k = 1; k <= numChans; k++) sum += fbank[k];
         *p++ = (sum * mfnorm) * c->cepScale;
      }
      if (c->hasE) {
         if (c->rawEnergy) te = rawte;
         *p++ = (te < ORC_MINLARG) ? ORC_LZERO : log(te);
      }
   }
   /* whole file: NormaliseLogEnergy HSigP.c:911 (HParm.c:4100-4103), then AddQualifiers HParm.c:1618 */
   if (c->hasE && c->eNormalise) {
      float *pp = out + nStat - 1, mx, mn;
      mx = *pp;
      for (i = 1; i < T; i++) { pp += nCols; if (*pp > mx) mx = *pp; }
      mn = mx - (c->silFloor * log(10.0)) / 10.0;
      pp = out + nStat - 1;
      for (i = 0; i < T; i++) {
         if (*pp < mn) *pp = mn;
         *pp = 1.0 - (mx - *pp) * c->eScale;
         pp += nCols;
      }
   }
   if (c->hasD) add_diffs(out, T, nCols, 0, nStat, nStat, c->delWin);
   if (c->hasA) add_diffs(out, T, nCols, nStat, 2 * nStat, nStat, c->accWin);
   if (c->hasZ) {                                                                     /* FZeroMean HSigP.c:803 */
      int d = c->numCeps + (c->hasC0 ? 1 : 0);
      for (i = 0; i < d; i++) {
         double sum = 0.0; float mean, *fp = out + i;
         for (j = 0; j < T; j++) { sum += *fp; fp += nCols; }
         mean = sum / (double)T;
         fp = out + i;
         for (j = 0; j < T; j++) { *fp -= mean; fp += nCols; }
      }
   }
   free(cf); free(loChan); free(loWt); free(ham); free(cepWin); free(s); free(x); free(fbank); free(cc);
   return T;
}

/* AddQualifiers on a parameterised table (HParm.c:1618): statics [T x nStat] -> [T x nStat*(1+D+A)] */
int orc_add_qualifiers(const float *stat, int T, int nStat, int hasD, int hasA, int delWin, int accWin, float *out)
{
   int nCols = nStat * (1 + (hasD ? 1 : 0) + (hasA ? 1 : 0)), t, k;
   for (t = 0; t < T; t++)
      for (k = 0; k < nStat; k++) out[(size_t)t * nCols + k] = stat[(size_t)t * nStat + k];
   if (T == 0) return nCols;
   if (hasD) add_diffs(out, T, nCols, 0, nStat, nStat, delWin);
   if (hasA) add_diffs(out, T, nCols, nStat, 2 * nStat, nStat, accWin);
   return nCols;
}

/* AddDiffs with the two variants of the difference computation (HParm.c:1552-1598): SIMPLEDIFFS = (c[t+w] - c[t-w]) / 2w with
   the same edge replication (Regress, HSigP.c:846-849), V1COMPAT = the first / last w rows are plain forward / backward
   differences (AddHeadRegress / AddTailRegress with delwin 0, HSigP.c:866-908); tables shorter than 2w+1 rows take the one-call
   regression in either case (HParm.c:1566-1573). */
static void regress_simple(float *data, int vSize, int n, int step, int offset, int delwin, int head, int tail)
{
   float *fp = data, *fp1, *fp2, *back, *forw;
   int i, t, j;
   for (i = 1; i <= n; i++) {
      fp1 = fp; fp2 = fp + offset;
      for (j = 1; j <= vSize; j++) {
         back = forw = fp1;
         for (t = 1; t <= delwin; t++) {
            if (head + i - t > 0) back -= step;
            if (tail + n - i + 1 - t > 0) forw += step;
         }
         *fp2 = (*forw - *back) / (2 * delwin);
         ++fp1; ++fp2;
      }
      fp += step;
   }
}

static void add_diffs_mode(float *data, int nRows, int nCols, int si, int ti, int d, int winSize, int v1Compat, int simpleDiffs)
{
   float *p = data + si;
   int offset = ti - si, head = winSize, tail = winSize, n = nRows - (head + tail), i, j;
   if (!v1Compat && !simpleDiffs) { add_diffs(data, nRows, nCols, si, ti, d, winSize); return; }
#define REG(pp, nn, hh, tt) do { if (simpleDiffs) regress_simple(pp, d, nn, nCols, offset, winSize, hh, tt); else regress(pp, d, nn, nCols, offset, winSize, hh, tt); } while (0)
   if (n <= 0) { REG(p, nRows, 0, 0); return; }
   if (v1Compat) {
      for (i = 0; i < head; i++) for (j = 0; j < d; j++) p[i * nCols + j + offset] = p[(i + 1) * nCols + j] - p[i * nCols + j];
   } else REG(p, head, 0, winSize);
   p += head * nCols;
   REG(p, n, winSize, winSize);
   p += n * nCols;
   if (v1Compat) {
      for (i = 0; i < tail; i++) for (j = 0; j < d; j++) p[i * nCols + j + offset] = p[i * nCols + j] - p[(i - 1) * nCols + j];
   } else REG(p, tail, winSize, 0);
#undef REG
}

/* The remaining qualifiers of AddQualifiers on a table: third differentials (HParm.c:1675-1681, regression of the
   accelerations over THIRDWINDOW), _Z after the differentials (HParm.c:1700-1726: FZeroMean over the first nZeroMean
   columns), and _N (the absolute energy / C0 column nullECol is left out when the row is handed out as an observation,
   ExtractObservation HParm.c:2882-2893).  out holds nStat*(1+D+A+T) - (nullECol >= 0) columns; returns that number. */
int orc_parm_qualify2(const float *stat, int T, int nStat, int nZeroMean, int hasD, int hasA, int hasT,
                      int delWin, int accWin, int thirdWin, int nullECol, int v1Compat, int simpleDiffs, float *out);
int orc_parm_qualify(const float *stat, int T, int nStat, int nZeroMean, int hasD, int hasA, int hasT,
                     int delWin, int accWin, int thirdWin, int nullECol, float *out)
{
   return orc_parm_qualify2(stat, T, nStat, nZeroMean, hasD, hasA, hasT, delWin, accWin, thirdWin, nullECol, 0, 0, out);
}

int orc_parm_qualify2(const float *stat, int T, int nStat, int nZeroMean, int hasD, int hasA, int hasT,
                      int delWin, int accWin, int thirdWin, int nullECol, int v1Compat, int simpleDiffs, float *out)
{
   int nFull = nStat * (1 + (hasD ? 1 : 0) + (hasA ? 1 : 0) + (hasT ? 1 : 0)), nCols = nFull - (nullECol >= 0 ? 1 : 0), t, k, i, j;
   float *full = (nullECol >= 0) ? (float *)malloc(sizeof(float) * (size_t)(T ? T : 1) * nFull) : out;
   for (t = 0; t < T; t++)
      for (k = 0; k < nStat; k++) full[(size_t)t * nFull + k] = stat[(size_t)t * nStat + k];
   if (T > 0) {
      if (hasD) add_diffs_mode(full, T, nFull, 0, nStat, nStat, delWin, v1Compat, simpleDiffs);
      if (hasA) add_diffs_mode(full, T, nFull, nStat, 2 * nStat, nStat, accWin, v1Compat, simpleDiffs);
      if (hasT) add_diffs_mode(full, T, nFull, 2 * nStat, 3 * nStat, nStat, thirdWin, v1Compat, simpleDiffs);
      for (i = 0; i < nZeroMean; i++) {                                              /* FZeroMean HSigP.c:803 */
         double sum = 0.0; float mean, *fp = full + i;
         for (j = 0; j < T; j++) { sum += *fp; fp += nFull; }
         mean = sum / (double)T;
         fp = full + i;
         for (j = 0; j < T; j++) { *fp -= mean; fp += nFull; }
      }
   }
   if (nullECol >= 0) {
      for (t = 0; t < T; t++)
         for (k = 0, j = 0; k < nFull; k++)
            if (k != nullECol) out[(size_t)t * nCols + j++] = full[(size_t)t * nFull + k];
      free(full);
   }
   return nCols;
}

/* HCompV's global statistics (HTKTools/HCompV.c): AccVar :392-411 adds every observation, in file order, into FLOAT
   accumulators (sum and sum of squares per component); CalcCovs :261-291 divides by the float frame count and floors the
   variance at minVar (default 0.0).  X: all frames [T x D] in the order the files were given. */
void orc_compv(const float *X, long T, int D, float minVar, float *mean, float *var)
{
   long t; int k;
   float *sum = (float *)calloc((size_t)D, sizeof(float)), *sq = (float *)calloc((size_t)D, sizeof(float));
   float n = (float)T;
   for (t = 0; t < T; t++)
      for (k = 0; k < D; k++) {
         float val = X[(size_t)t * D + k];
         sum[k] += val;
         sq[k] += val * val;
      }
   for (k = 0; k < D; k++) mean[k] = sum[k] / n;
   for (k = 0; k < D; k++) {
      float meanx = mean[k], varxy = sq[k] / n - meanx * meanx;
      var[k] = (varxy > minVar) ? varxy : minVar;
   }
   free(sum); free(sq);
}
