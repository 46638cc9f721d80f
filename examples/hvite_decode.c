/* hvite_decode.c -- recognition of a set of parameter files over a word network (HVite -w) against the C ABI alone:
 *
 *     hvite_decode <hmmList> <hmmDir> <lattice.slf> <dict> <outDir> <genBeam> <lmScale> <wordPen> file1.mfc ...
 *
 * loads the models, expands the SLF lattice + dictionary into the recognition network, appends deltas on the device when the
 * models' kind asks for them, decodes all files as one batch and writes <outDir>/<base>.rec with the reference's label lines
 * ("start end symbol score", 100 ns units).  tests/test_gpu_demo.py builds it with gcc and compares its files with HVite's.
 *
 *     gcc -O2 -Iinclude examples/hvite_decode.c -o hvite_decode -Lhtk_amd -lhtk_amd -Wl,-rpath,$PWD/htk_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "htk_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, htkamd_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
   if (argc < 10) { fprintf(stderr, "usage: %s hmmList hmmDir lattice dict outDir genBeam lmScale wordPen files...\n", argv[0]); return 2; }
   const char *hmmList = argv[1], *hmmDir = argv[2], *slf = argv[3], *dict = argv[4], *outDir = argv[5];
   const float genBeam = (float)atof(argv[6]), lmScale = (float)atof(argv[7]), wordPen = (float)atof(argv[8]);
   const int nFiles = argc - 9;
   char **files = argv + 9;

   htkamd_mmf *mmf; CHECK(htkamd_mmf_create(&mmf));
   CHECK(htkamd_mmf_finish(mmf, hmmList, hmmDir, NULL));
   const htkamd_model_desc *d = htkamd_mmf_desc(mmf);
   const char *kind = htkamd_mmf_parm_kind(mmf);
   const int hasD = strstr(kind, "_D") != NULL, hasA = strstr(kind, "_A") != NULL;
   htkamd_model *model; CHECK(htkamd_model_create(d, &model));
   htkamd_net *net; CHECK(htkamd_net_build(slf, dict, mmf, &net));
   htkamd_decoder *dec; CHECK(htkamd_decoder_create(model, htkamd_net_get(net), lmScale, &dec));

   int *frameOff = (int *)calloc((size_t)nFiles + 1, sizeof(int));
   float *stat = NULL; int nStat = 0, period = 100000;
   for (int u = 0; u < nFiles; u++) {
      float *x; int T, cols, pk;
      CHECK(htkamd_parm_read(files[u], &x, &T, &cols, &period, &pk));
      if (u == 0) nStat = cols;
      if (cols != nStat) { fprintf(stderr, "%s: %d columns, expected %d\n", files[u], cols, nStat); return 1; }
      stat = (float *)realloc(stat, sizeof(float) * (size_t)(frameOff[u] + T) * nStat);
      memcpy(stat + (size_t)frameOff[u] * nStat, x, sizeof(float) * (size_t)T * nStat);
      frameOff[u + 1] = frameOff[u] + T;
      htkamd_free(x);
   }
   const int F = frameOff[nFiles], D = d->vecSize;
   if (nStat * (1 + hasD + hasA) != D) { fprintf(stderr, "files have %d statics, models want %d (%s)\n", nStat, D, kind); return 1; }
   float *dStat, *dX;
   CHECK(htkamd_dev_malloc((void **)&dStat, sizeof(float) * (size_t)F * nStat));
   CHECK(htkamd_dev_malloc((void **)&dX, sizeof(float) * (size_t)F * D));
   CHECK(htkamd_memcpy_h2d(dStat, stat, sizeof(float) * (size_t)F * nStat, NULL));
   CHECK(htkamd_parm_add_qualifiers(dStat, frameOff, nFiles, nStat, hasD, hasA, 2, 2, dX, NULL));

   const int maxWords = 4096;
   int *nWords = (int *)malloc(sizeof(int) * (size_t)nFiles), *wPron = (int *)malloc(sizeof(int) * (size_t)nFiles * maxWords);
   int *wStart = (int *)malloc(sizeof(int) * (size_t)nFiles * maxWords), *wEnd = (int *)malloc(sizeof(int) * (size_t)nFiles * maxWords);
   float *wScore = (float *)malloc(sizeof(float) * (size_t)nFiles * maxWords), *wLm = (float *)malloc(sizeof(float) * (size_t)nFiles * maxWords);
   double *total = (double *)malloc(sizeof(double) * (size_t)nFiles);
   htkamd_decode_config cfg; memset(&cfg, 0, sizeof(cfg));
   cfg.genBeam = genBeam > 0 ? genBeam : 1.0e10f; cfg.wordBeam = 1.0e10f; cfg.lmScale = lmScale; cfg.wordPen = wordPen; cfg.prScale = 1.0f;
   cfg.scoreMode = HTKAMD_SCORE_EXACT;
   CHECK(htkamd_decoder_run(dec, &cfg, dX, frameOff, nFiles, maxWords, nWords, wPron, wStart, wEnd, wScore, wLm, total, NULL));

   for (int u = 0; u < nFiles; u++) {
      char base[512], path[1024];
      const char *sl = strrchr(files[u], '/');
      snprintf(base, sizeof(base), "%s", sl ? sl + 1 : files[u]);
      char *dot = strrchr(base, '.'); if (dot) *dot = 0;
      snprintf(path, sizeof(path), "%s/%s.rec", outDir, base);
      FILE *f = fopen(path, "w");
      if (!f) { fprintf(stderr, "cannot create %s\n", path); return 1; }
      for (int w = 0; w < nWords[u]; w++) {
         const int k = u * maxWords + w;
         const char *sym = htkamd_net_out_sym(net, wPron[k]);
         if (!sym || !sym[0]) continue;                        /* words without output symbol leave no label (HRec.c:2342-2356) */
         fprintf(f, "%lld %lld %s %f\n", (long long)wStart[k] * period, (long long)wEnd[k] * period, sym, wScore[k]);
      }
      fclose(f);
      printf("%s: %d words, log probability %f\n", base, nWords[u], total[u]);
   }
   htkamd_decoder_destroy(dec); htkamd_net_destroy(net); htkamd_model_destroy(model); htkamd_mmf_destroy(mmf);
   htkamd_dev_free(dStat); htkamd_dev_free(dX);
   return 0;
}
