/* prep.c -- host-side (C) preparation of an HMM set for the device path.
 *
 * These are the once-per-load conversions HERest/HVite apply to an HMMSet before the hot path
 * starts; they stay on the host exactly as in the reference (they run once, in milliseconds):
 *   gConst      FixDiagGConst   HModel.c:5641-5654
 *   1/variance  ConvDiagC       HUtil.c:413-437
 *   log weight  MixLogWeight    HModel.c:5288-5295 (via ConvLogWt HUtil.c:474)
 *   min duration of a transition matrix  SetMinDurs  HFB.c:106-155
 * Compile without FMA contraction so that float results equal the reference's SSE2 arithmetic.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

double htkamd_host_min_log_exp(void)
{
   return -log(-LZERO);                 /* InitMath: minLogExp, HMath.c:1680 */
}

void htkamd_host_fix_diag_gconst(int D, const float *var, float *gconst)
{
   float sum = D * log(HTK_TPI);        /* double product rounded to float, as the reference */
   int i;
   for (i = 0; i < D; i++) {
      float z = (var[i] <= MINLARG) ? LZERO : log(var[i]);
      sum += z;
   }
   *gconst = sum;
}

void htkamd_host_conv_diagc(size_t n, const float *var, float *ivar)
{
   size_t k;
   for (k = 0; k < n; k++) {
      float v = var[k];
      if (v > 1E+30) v = 1E+30;
      if (v < 1E-30) v = 1E-30;
      ivar[k] = 1 / v;
   }
}

float htkamd_host_mix_log_weight(float w)
{
   return (w < MINMIX) ? LZERO : log(w);
}

/* Topological minimum number of frames needed to traverse a model (entry -> exit).
   tp: row-major N*N log transition matrix, state (i,j) 1-based at tp[(i-1)*N + j-1]. */
static void order_states(int N, const float *tp, int *mark, int s, int *count)
{
   int p;
   mark[s] = 0;
   for (p = 1; p < N; p++)
      if (p != s && tp[(p - 1) * N + (s - 1)] > LSMALL && mark[p] < 0)
         order_states(N, tp, mark, p, count);
   mark[s] = ++(*count);
}

int htkamd_host_min_dur(int N, const float *tp)
{
   int *rank = (int *)malloc(sizeof(int) * (size_t)(N + 1));
   int *byRank = (int *)malloc(sizeof(int) * (size_t)(N + 1));
   int *md = (int *)malloc(sizeof(int) * (size_t)(N + 1));
   int i, j, k, n = 0, res;
   for (i = 1; i <= N; i++) { rank[i] = -1; byRank[i] = -1; }
   order_states(N, tp, rank, N, &n);            /* depth-first from the exit state over predecessors */
   for (i = 1; i <= N; i++)
      if (rank[i] >= 1) byRank[rank[i]] = i;
   for (i = 1; i <= N; i++) md[i] = N;
   md[1] = 0;
   for (k = 1; k <= n; k++) {
      i = byRank[k];
      if (i < 1 || i > N) continue;
      for (j = 1; j < N; j++)
         if (tp[(j - 1) * N + (i - 1)] > LSMALL) {
            int d = md[j] + ((i == N) ? 0 : 1);
            if (d < md[i]) md[i] = d;
         }
   }
   if (md[N] < 0 || md[N] >= N)
      res = (tp[N - 1] > LSMALL) ? 0 : 1;       /* discontinuous matrix: under-estimate (HFB.c:144-149) */
   else
      res = md[N];
   free(rank); free(byRank); free(md);
   return res;
}


/* ------------------------------------------------------------------------------------------
 * Table for the device LAdd.  LAdd(x,y) = x + log(1 + exp(y-x)) (HMath.c:1576-1590) is evaluated
 * on the device as x + f(d), d = y-x in [minLogExp, 0], f(d) = log1p(exp(d)), with f taken from a
 * piecewise Taylor polynomial: interval k = (int)(-d*LADD_INV_H) has centre c_k = -(k+0.5)/LADD_INV_H
 * and f(c_k + r) = sum_n tab[k][n] r^n, |r| <= 1/(2*LADD_INV_H).  With 4 intervals per unit and
 * degree 10 the truncation error is below 1e-17, i.e. at the rounding level of the glibc
 * exp()/log() pair the reference calls, so float-rounded results agree except when the double
 * result sits within ~1e-16 of a float rounding boundary.
 * Derivatives: f' = s, s = 1/(1+exp(-d));  f^(n+1) = (d/ds f^(n)) * s(1-s), polynomials in s.
 * ------------------------------------------------------------------------------------------ */
int htkamd_host_ladd_table_size(void) { return LADD_NK * (LADD_DEG + 1); }

void htkamd_host_build_ladd_table(double *tab)
{
   /* polynomials P_n(s) with f^(n)(d) = P_n(s(d)), n = 1..LADD_DEG */
   long double P[LADD_DEG + 1][LADD_DEG + 2];
   int n, i, k;
   for (n = 0; n <= LADD_DEG; n++)
      for (i = 0; i <= LADD_DEG + 1; i++) P[n][i] = 0.0L;
   P[1][1] = 1.0L;                                           /* f' = s */
   for (n = 1; n < LADD_DEG; n++) {
      /* P_{n+1} = P_n'(s) * (s - s^2) */
      long double dP[LADD_DEG + 2];
      for (i = 0; i <= LADD_DEG; i++) dP[i] = (i + 1) * P[n][i + 1];
      dP[LADD_DEG + 1] = 0.0L;
      for (i = 0; i <= LADD_DEG + 1; i++) {
         long double v = 0.0L;
         if (i >= 1) v += dP[i - 1];
         if (i >= 2) v -= dP[i - 2];
         P[n + 1][i] = v;
      }
   }
   for (k = 0; k < LADD_NK; k++) {
      long double c = -((long double)k + 0.5L) / (long double)LADD_INV_H;
      long double s = 1.0L / (1.0L + expl(-c));
      long double fact = 1.0L;
      double *row = tab + (size_t)k * (LADD_DEG + 1);
      row[0] = (double)log1pl(expl(c));
      for (n = 1; n <= LADD_DEG; n++) {
         long double v = 0.0L, sp = 1.0L;
         fact *= n;
         for (i = 0; i <= LADD_DEG + 1; i++) { v += P[n][i] * sp; sp *= s; }
         row[n] = (double)(v / fact);
      }
   }
}

/* Left-to-right without skips (fb_lr.hip): the entry state reaches state 2 only, emitting state i reaches i and i+1 only, the exit
   state is reached from state N-1 only; no tee transition.  Transitions inside the pattern may be log-zero. */
int htkamd_host_trans_is_lr(int N, const float *tp)
{
   if (N < 3 || N > 5) return 0;
   for (int j = 3; j <= N; j++) if (tp[j - 1] > (float)LSMALL) return 0;                   /* a_1j, j > 2 (j = N: tee) */
   for (int i = 2; i <= N - 1; i++)
      for (int j = 2; j <= N; j++) {
         const int ok = (j == i) || (j == i + 1);                                           /* i+1 == N for the last emitting state */
         if (!ok && tp[(i - 1) * N + (j - 1)] > (float)LSMALL) return 0;
      }
   return 1;
}


/* ---- several streams --------------------------------------------------------------------------------------------------
 * A multi-stream set splits the observation vector into S stream vectors: SetStreamWidths (HParm.c:3094-3168: the standard split of
 * the parameter kind, `eSep` = the energy terms form the last stream) and ExtractObservation (HParm.c:2843-2895: where each element of
 * the table row goes).  Here the split is a map dimension -> stream over the undivided row (dimStream), each stream's elements in
 * ascending order -- the order ExtractObservation fills a stream vector in, so sums over a stream's dimensions run in the reference's
 * order.  kind: the target kind as text ("MFCC_E_D"); width[s]: <STREAMINFO>.  Returns 0, or -1 (no such split: message in `why`). */
int htkamd_host_stream_dims(const char *kind, int vecSize, int S, const int *width, int *dimStream, char *why, size_t whyLen)
{
   int hasE = 0, has0 = 0, hasN = 0, hasD = 0, hasA = 0, hasT = 0, s, k, tot = 0;
   const char *p = kind ? strchr(kind, '_') : NULL;
   if (S < 1 || S > 7) { snprintf(why, whyLen, "%d streams (1..7 supported)", S); return -1; }
   for (s = 0; s < S; s++) { if (width[s] < 1) { snprintf(why, whyLen, "stream %d has width %d", s + 1, width[s]); return -1; } tot += width[s]; }
   if (tot != vecSize) { snprintf(why, whyLen, "stream widths sum to %d, vector size is %d", tot, vecSize); return -1; }
   for (; p && *p; p++) {
      if (*p != '_') continue;
      switch (p[1]) { case 'E': hasE = 1; break; case '0': has0 = 1; break; case 'N': hasN = 1; break; case 'D': hasD = 1; break; case 'A': hasA = 1; break; case 'T': hasT = 1; break; default: break; }
   }
   if (S == 1) { for (k = 0; k < vecSize; k++) dimStream[k] = 0; return 0; }
   if (hasN) { snprintf(why, whyLen, "several streams with the _N qualifier are not supported"); return -1; }
   /* the standard split, if there is one (SetStreamWidths) */
   {
      const int nBlocks = 1 + hasD + (hasD && hasA) + (hasD && hasA && hasT);
      const int en = (hasE || has0) ? 1 : 0;                            /* NumEnergy counts one term per block */
      const int neObs = en ? nBlocks : 0;
      int sw[8] = {0}, ok = 1, eSep = 0;
      if ((hasE && has0) || vecSize % nBlocks) ok = 0;
      const int blk = vecSize / nBlocks, stat = blk - en;
      if (ok) switch (S) {
         case 2:
            if (en) { sw[1] = neObs; sw[0] = vecSize - neObs; eSep = 1; }
            else if (!hasA && hasD) { sw[1] = blk; sw[0] = vecSize - blk; }
            else ok = 0;
            break;
         case 3:
            if (hasA) { sw[1] = blk; sw[2] = blk; sw[0] = vecSize - 2 * blk; }
            else if (hasD && en) { sw[0] = sw[1] = stat; sw[2] = neObs; eSep = 1; }
            else ok = 0;
            break;
         case 4:
            if (hasA && en) { sw[0] = sw[1] = sw[2] = stat; sw[3] = neObs; eSep = 1; }
            else ok = 0;
            break;
         default: ok = 0; break;
      }
      if (ok) for (s = 0; s < S; s++) if (sw[s] != width[s]) { eSep = 0; break; }
      if (eSep) {
         /* the row is nBlocks blocks of (stat coefficients, energy): ExtractObservation :2855-2877 */
         for (k = 0; k < vecSize; k++) {
            const int b = k / blk, i = k % blk;
            if (i == blk - 1) dimStream[k] = S - 1;
            else dimStream[k] = (S == 2) ? 0 : b;
         }
         return 0;
      }
   }
   /* any other split: the streams are consecutive pieces of the row (ExtractObservation :2883-2893) */
   for (s = 0, k = 0; s < S; s++) { int j; for (j = 0; j < width[s]; j++) dimStream[k++] = s; }
   return 0;
}

/* FixDiagGConst (HModel.c:5641) for the Gaussian of one stream held in an undivided row: n log(2 pi) + sum of log variances over the
   stream's dimensions in ascending order.  dimStream == NULL: every dimension. */
void htkamd_host_fix_diag_gconst_ms(int D, const float *var, const int *dimStream, int stream, float *gconst)
{
   int i, n = 0;
   float sum;
   for (i = 0; i < D; i++) if (!dimStream || dimStream[i] == stream) n++;
   sum = n * log(HTK_TPI);
   for (i = 0; i < D; i++)
      if (!dimStream || dimStream[i] == stream) {
         float z = (var[i] <= MINLARG) ? LZERO : log(var[i]);
         sum += z;
      }
   *gconst = sum;
}
