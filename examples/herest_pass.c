/* herest_pass.c -- one embedded Baum-Welch pass (HERest) written against the C ABI alone (include/htk_amd.h):
 *
 *     herest_pass <hmmList> <hmmDir> <labDir> <outDir> <pruneThresh> <minVar> <mixFloorMult> file1.mfc file2.mfc ...
 *
 * reads the model files of <hmmDir> named in <hmmList>, the parameter files (statics, e.g. MFCC_E; deltas are appended on the
 * device when the models' kind has _D / _A), the label files <labDir>/<base>.lab; runs scoring, forward-backward and the
 * statistics on the GPU, UpdateModels on the host, writes the models to <outDir> and prints HERest's summary line.
 * This is what a C host (the reference's HERest.c with its HTKLib calls swapped for the library's) does; tests/test_gpu_demo.py
 * builds it with gcc and runs it on the HTKDemo fixtures.
 *
 *     gcc -O2 -Iinclude examples/herest_pass.c -o herest_pass -Lhtk_amd -lhtk_amd -Wl,-rpath,$PWD/htk_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "htk_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, htkamd_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
   if (argc < 9) { fprintf(stderr, "usage: %s hmmList hmmDir labDir outDir pruneThresh minVar mixFloorMult files...\n", argv[0]); return 2; }
   const char *hmmList = argv[1], *hmmDir = argv[2], *labDir = argv[3], *outDir = argv[4];
   const double prune = atof(argv[5]);
   const float minVar = (float)atof(argv[6]), mixFloor = (float)(atof(argv[7]) * 1.0e-5);       /* HERest -w f = f * MINMIX */
   const int nFiles = argc - 8;
   char **files = argv + 8;

   /* LoadHMMSet */
   htkamd_mmf *mmf; CHECK(htkamd_mmf_create(&mmf));
   CHECK(htkamd_mmf_finish(mmf, hmmList, hmmDir, NULL));
   const htkamd_model_desc *d = htkamd_mmf_desc(mmf);
   const char *kind = htkamd_mmf_parm_kind(mmf);
   const int hasD = strstr(kind, "_D") != NULL, hasA = strstr(kind, "_A") != NULL;
   htkamd_model *model; CHECK(htkamd_model_create(d, &model));

   /* parameter files + label files -> one table of statics, frame offsets, model sequences */
   int *frameOff = (int *)calloc((size_t)nFiles + 1, sizeof(int)), *labOff = (int *)calloc((size_t)nFiles + 1, sizeof(int));
   float *stat = NULL; int *labs = NULL; int nStat = 0, capLab = 0;
   for (int u = 0; u < nFiles; u++) {
      float *x; int T, cols, period, pk;
      CHECK(htkamd_parm_read(files[u], &x, &T, &cols, &period, &pk));
      if (u == 0) nStat = cols;
      if (cols != nStat) { fprintf(stderr, "%s: %d columns, expected %d\n", files[u], cols, nStat); return 1; }
      stat = (float *)realloc(stat, sizeof(float) * (size_t)(frameOff[u] + T) * nStat);
      memcpy(stat + (size_t)frameOff[u] * nStat, x, sizeof(float) * (size_t)T * nStat);
      frameOff[u + 1] = frameOff[u] + T;
      htkamd_free(x);
      char base[512], lab[1024];
      const char *sl = strrchr(files[u], '/');
      snprintf(base, sizeof(base), "%s", sl ? sl + 1 : files[u]);
      char *dot = strrchr(base, '.'); if (dot) *dot = 0;
      snprintf(lab, sizeof(lab), "%s/%s.lab", labDir, base);
      htkamd_labels *L; CHECK(htkamd_labels_read(lab, &L));
      const int n = htkamd_labels_count(L);
      if (labOff[u] + n > capLab) { capLab = (labOff[u] + n) * 2 + 64; labs = (int *)realloc(labs, sizeof(int) * (size_t)capLab); }
      for (int i = 0; i < n; i++) {
         const int h = htkamd_mmf_find_logical(mmf, htkamd_labels_name(L, i));
         if (h < 0) { fprintf(stderr, "%s: no model %s\n", lab, htkamd_labels_name(L, i)); return 1; }
         labs[labOff[u] + i] = h;
      }
      labOff[u + 1] = labOff[u] + n;
      htkamd_labels_free(L);
   }
   const int F = frameOff[nFiles], D = d->vecSize;
   if (nStat * (1 + hasD + hasA) != D) { fprintf(stderr, "files have %d statics, models want %d (%s)\n", nStat, D, kind); return 1; }
   float *dStat, *dX;
   CHECK(htkamd_dev_malloc((void **)&dStat, sizeof(float) * (size_t)F * nStat));
   CHECK(htkamd_dev_malloc((void **)&dX, sizeof(float) * (size_t)F * D));
   CHECK(htkamd_memcpy_h2d(dStat, stat, sizeof(float) * (size_t)F * nStat, NULL));
   CHECK(htkamd_parm_add_qualifiers(dStat, frameOff, nFiles, nStat, hasD, hasA, 2, 2, dX, NULL));

   /* the pass */
   htkamd_accs *accs; CHECK(htkamd_accs_create(model, &accs));
   htkamd_fb *fb; CHECK(htkamd_fb_create(model, &fb));
   htkamd_batch_desc b = { nFiles, dX, frameOff, labOff, labs };
   htkamd_fb_config cfg; memset(&cfg, 0, sizeof(cfg));
   cfg.pruneInit = cfg.pruneLim = prune > 0 ? prune : HTKAMD_NOPRUNE; cfg.pruneInc = 0.0;
   cfg.minFrwdP = 10.0f; cfg.uFlags = HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES | HTKAMD_UPTRANS; cfg.scoreMode = HTKAMD_SCORE_EXACT;
   CHECK(htkamd_accs_zero(accs, NULL));
   CHECK(htkamd_fb_prepare(fb, &b, NULL));
   CHECK(htkamd_fb_execute(fb, &cfg, accs, NULL));
   double *pr = (double *)malloc(sizeof(double) * (size_t)nFiles); int *st = (int *)malloc(sizeof(int) * (size_t)nFiles);
   CHECK(htkamd_fb_results(fb, pr, st, NULL));
   htkamd_accs_layout lay; CHECK(htkamd_accs_get_layout(accs, &lay));
   double *vec = (double *)malloc(sizeof(double) * lay.total);
   CHECK(htkamd_accs_download(accs, vec, NULL));

   /* UpdateModels + SaveHMMSet */
   htkamd_update_config uc; memset(&uc, 0, sizeof(uc));
   uc.minEgs = 3; uc.minVar = minVar; uc.mixWeightFloor = mixFloor; uc.uFlags = cfg.uFlags; uc.varFloor = htkamd_mmf_var_floor(mmf);
   htkamd_update_stats us;
   CHECK(htkamd_model_update(model, accs, vec, &uc, &us));
   float *mean = (float *)malloc(sizeof(float) * (size_t)d->numGauss * D), *var = (float *)malloc(sizeof(float) * (size_t)d->numGauss * D);
   float *gc = (float *)malloc(sizeof(float) * (size_t)d->numGauss), *wt = (float *)malloc(sizeof(float) * (size_t)d->numComp);
   float *tp = (float *)malloc(sizeof(float) * (size_t)d->transOff[d->numTrans]);
   CHECK(htkamd_model_get_params(model, mean, var, gc, wt, tp));
   CHECK(htkamd_mmf_write(mmf, mean, var, gc, wt, tp, NULL, outDir));
   if (us.nFloorVar > 0) printf("Total %d floored variance elements in %d different mixes\n", us.nFloorVar, us.nFloorVarMix);
   printf("Reestimation complete - average log prob per frame = %e\n", vec[lay.totalPr] / vec[lay.totalT]);
   printf("     - total frames seen          = %e\n", vec[lay.totalT]);
   htkamd_fb_destroy(fb); htkamd_accs_destroy(accs); htkamd_model_destroy(model); htkamd_mmf_destroy(mmf);
   htkamd_dev_free(dStat); htkamd_dev_free(dX);
   return 0;
}
