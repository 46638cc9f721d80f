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
#include <stdlib.h>
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
