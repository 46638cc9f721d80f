/* herest.c -- embedded Baum-Welch re-estimation on the MI355X: the command-line program a user of the reference's HERest switches to.
 *
 *   herest [options] hmmList dataFiles...
 *
 * Options (the subset of HTKBook ref.tex "HERest" that SURVEY.md 8(b) lists; same letters, same meaning, same defaults, HERest.c:292-500):
 *   -C cf          configuration file (TARGETKIND, DELTAWINDOW, ACCWINDOW, THIRDWINDOW, V1COMPAT, SIMPLEDIFFS)
 *   -S scp         script file: further data files (extended file names logical=physical[s,e] are accepted, the segment is ignored)
 *   -H mmf         load a master macro file (repeatable)          -d dir / -x ext   directory / extension of single-model files
 *   -M dir         directory for the re-estimated models          -B                save them in binary
 *   -L dir / -X ext  directory / extension of the label files (default: beside the data, .lab)      -I mlf   master label file
 *   -t f [i l]     beam pruning threshold [increment limit]       -c f              mixture pruning threshold (MINFORPROB, default 10)
 *   -u tmvw        parameters to update (default tmvw)            -m N              minimum examples per model (3)
 *   -v f           variance floor (0)                             -w f              mixture weight floor as a multiple of MINMIX (0)
 *   -p N           parallel mode: N > 0 accumulates this shard and writes <-M dir>/HERN.acc; N = 0 takes the data arguments as accumulator
 *                  files, sums them and re-estimates (HERest.c:502-557)
 *   -s file        write the state occupation statistics file     -T N              trace (1: progress lines)
 * Beyond the reference:
 *   --score m      arithmetic: exact (default: alpha/beta/scores bit-identical to the reference), fast (fp32 matrix-core scores +
 *                  fp32-transcendental LAdd), bf16 (bf16 x 3 matrix-core scores + that LAdd), fastest (fp16 x 2 matrix-core scores + that
 *                  LAdd; an iteration whose data or model does not fit fp16's range -- HTKAMD_ERANGE -- is repeated as bf16, with a
 *                  warning); all but exact are tolerance class (1e-4)
 *   --batch N      utterances per device batch (default 4096)
 *   --iterations K  K Baum-Welch iterations in one process: features, transcriptions and batch tables stay on the device, the model is
 *                  re-estimated where it is (htkamd_model_update_device) and only the last iteration's set is written to -M (HTK's recipe runs
 *                  one HERest process per iteration: load 88 MB, save 88 MB around 5 ms of device work)
 *   --ranks R --rank r --rccl-id file   one process per GPU: every rank takes the data files r, r+R, r+2R, ..., the accumulators are summed
 *                  over RCCL (htkamd_accs_allreduce) and every rank re-estimates; rank 0 writes the models.  `file` carries the
 *                  rendezvous id from rank 0 to the others, tagged with the run's nonce (--rccl-nonce N; default: the launcher's process
 *                  id, which the ranks of one run share); rank 0 removes the file before writing and at exit.  --rccl-timeout S (120):
 *                  a rank that cannot meet the others within S seconds -- at the rendezvous or in the all-reduce -- exits with status 3.
 *                  --wire f32 | f64: the statistics on the wire as floats (default: what HERest's own accumulator dumps carry, HTrain.c:1453-1505,
 *                  and what bench.py's multi-GPU line is measured with; every rank rounds its fp64 partial sums once, counters stay fp64) or as doubles.
 * Output on stdout follows HERest -T 1: "Pruning-On[..]", a line per skipped file, "Total N floored variance elements ...",
 * "Reestimation complete - average log prob per frame = ...".
 */
#include <signal.h>
#include <unistd.h>
#include "cli_common.h"

typedef struct { char *logical; int phys; } modelref;

#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
/* -T 16384 (octal 040000, a bit HERest.c does not use: its T_TOP / T_MAP / T_UPD are 1 / 2 / 4): wall-clock seconds per phase of the run, on stdout at the end */
static double g_t[8]; static const char *const g_tn[8] = {"load model set", "create device model", "read data + labels", "prepare", "execute + results", "update", "save model set", "other"};
#define TIC double tic_ = now_s()
#define TOC(k) do { const double n_ = now_s(); g_t[k] += n_ - tic_; tic_ = n_; } while (0)

static int uflags_parse(const char *s)
{
   int f = 0;
   for (; *s; s++)
      switch (*s) {
      case 't': f |= HTKAMD_UPTRANS; break; case 'm': f |= HTKAMD_UPMEANS; break; case 'v': f |= HTKAMD_UPVARS; break; case 'w': f |= HTKAMD_UPMIXES; break;
      case 'p': f |= HTKAMD_UPMAP; break;
      default: DIE("-u: unknown update flag %c (t m v w p)", *s);
      }
   return f;
}

static const char *g_rcclIdFile = NULL;
static int g_rank = 0;
static void remove_id_file(void) { if (g_rcclIdFile) unlink(g_rcclIdFile); }
static void rendezvous_timeout(int sig)
{
   (void)sig;
   static const char msg[] = "ERROR herest: the ranks did not meet within --rccl-timeout seconds (a rank missing or dead)\n";
   if (write(2, msg, sizeof(msg) - 1) < 0) { }
   if (g_rank == 0 && g_rcclIdFile) unlink(g_rcclIdFile);
   _exit(3);
}

int main(int argc, char **argv)
{
   args a = {argc, 1, argv};
   config cfg; memset(&cfg, 0, sizeof(cfg));
   strlist mmfs = {0}, files = {0};
   const char *hmmDir = NULL, *hmmExt = NULL, *outDir = NULL, *labDir = NULL, *labExt = "lab", *mlfPath = NULL, *statsFile = NULL, *rcclIdFile = NULL;
   double pruneInit = HTKAMD_NOPRUNE, pruneInc = 0.0, pruneLim = HTKAMD_NOPRUNE;
   float minFrwdP = 10.0f, minVar = 0.0f, mixFloor = 0.0f;
   int uFlags = HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES | HTKAMD_UPTRANS, minEgs = 3, parMode = -1, trace = 0, binary = 0;
   int scoreMode = HTKAMD_SCORE_EXACT, batchN = 4096, nRanks = 1, rank = 0, rcclTimeout = 120, nIter = 1, wire = HTKAMD_WIRE_F32, compat = 0;
   unsigned long long rcclNonce = (unsigned long long)getppid();      /* the ranks of one run are children of one launcher; --rccl-nonce overrides */
   const char *sw;

   while (a.at < a.argc && (a.argv[a.at][0] == '-') && (is_switch(a.argv[a.at]))) {
      if (!strncmp(a.argv[a.at], "--", 2)) {
         const char *lo = a.argv[a.at++] + 2;
         if (!strcmp(lo, "score")) {
            const char *m = str_arg(&a, "-score");
            scoreMode = !strcmp(m, "exact") ? HTKAMD_SCORE_EXACT : !strcmp(m, "fast") ? HTKAMD_SCORE_FAST : !strcmp(m, "fastest") ? HTKAMD_SCORE_FASTEST :
                        !strcmp(m, "bf16") ? (HTKAMD_SCORE_BF16 | HTKAMD_SCORE_FASTLADD) : -1;
            if (scoreMode < 0) DIE("--score: exact | fast | bf16 | fastest");
         } else if (!strcmp(lo, "batch")) batchN = atoi(str_arg(&a, "-batch"));
         else if (!strcmp(lo, "ranks")) nRanks = atoi(str_arg(&a, "-ranks"));
         else if (!strcmp(lo, "rank")) rank = atoi(str_arg(&a, "-rank"));
         else if (!strcmp(lo, "iterations")) nIter = atoi(str_arg(&a, "-iterations"));
         else if (!strcmp(lo, "rccl-id")) rcclIdFile = str_arg(&a, "-rccl-id");
         else if (!strcmp(lo, "rccl-nonce")) rcclNonce = strtoull(str_arg(&a, "-rccl-nonce"), NULL, 0);
         else if (!strcmp(lo, "rccl-timeout")) rcclTimeout = atoi(str_arg(&a, "-rccl-timeout"));
         else if (!strcmp(lo, "wire")) {
            const char *m = str_arg(&a, "-wire");
            wire = !strcmp(m, "f32") ? HTKAMD_WIRE_F32 : !strcmp(m, "f64") ? HTKAMD_WIRE_F64 : -1;
            if (wire < 0) DIE("--wire: f32 | f64");
         }
         else if (!strcmp(lo, "compat")) compat = HTKAMD_COMPAT_STREAM_REVISIT | HTKAMD_COMPAT_SHARED_LOGWT;      /* HERest's own numbers on sets of 2 or 4+ streams (HFB.c:1059) and on sets with shared pdfs (HUtil.c:474) */
         else DIE("unknown option --%s", lo);
         continue;
      }
      sw = next_switch(&a);
      switch (sw[0]) {
      case 'C': cfg_read(&cfg, str_arg(&a, sw)); break;
      case 'S': {
         htkamd_scp *scp; CHECK(htkamd_scp_read(str_arg(&a, sw), &scp));
         for (int i = 0; i < htkamd_scp_count(scp); i++) sl_add(&files, htkamd_scp_physical(scp, i));
         htkamd_scp_free(scp);
         break;
      }
      case 'H': sl_add(&mmfs, str_arg(&a, sw)); break;
      case 'd': hmmDir = str_arg(&a, sw); break;
      case 'x': hmmExt = str_arg(&a, sw); break;
      case 'M': outDir = str_arg(&a, sw); break;
      case 'B': binary = 1; break;
      case 'L': labDir = str_arg(&a, sw); break;
      case 'X': labExt = str_arg(&a, sw); break;
      case 'I': mlfPath = str_arg(&a, sw); break;
      case 't':
         pruneInit = flt_arg(&a, sw);
         if (has_num_arg(&a)) { pruneInc = flt_arg(&a, sw); pruneLim = flt_arg(&a, sw); } else { pruneInc = 0.0; pruneLim = pruneInit; }
         break;
      case 'c': minFrwdP = (float)flt_arg(&a, sw); break;
      case 'u': uFlags = uflags_parse(str_arg(&a, sw)); break;
      case 'm': minEgs = atoi(str_arg(&a, sw)); break;
      case 'v': minVar = (float)flt_arg(&a, sw); break;
      case 'w': mixFloor = (float)(1.0e-5 * flt_arg(&a, sw)); break;          /* MINMIX * f (HERest.c:425) */
      case 'p': parMode = atoi(str_arg(&a, sw)); break;
      case 's': statsFile = str_arg(&a, sw); break;
      case 'T': trace = atoi(str_arg(&a, sw)); break;
      default: DIE("herest: unknown switch -%s", sw);
      }
   }
   if (a.at >= a.argc) DIE("herest: file name of the HMM list expected");
   const char *hmmList = a.argv[a.at++];
   while (a.at < a.argc) sl_add(&files, a.argv[a.at++]);
   if (files.n == 0) DIE("herest: no data files");
   if (nRanks < 1 || rank < 0 || rank >= nRanks) DIE("herest: --rank must be in 0..ranks-1");
   if (nRanks > 1 && !rcclIdFile) DIE("herest: --ranks needs --rccl-id <file>");
   if (nIter < 1) DIE("herest: --iterations must be at least 1");
   if (nIter > 1 && parMode >= 0) DIE("herest: --iterations goes with the single-process form (or --ranks), not with -p");

   if (htkamd_device_count() <= 0) DIE("herest: no HIP device (the MI355X path has no CPU fallback)");
   CHECK(htkamd_set_device(nRanks > 1 ? rank % htkamd_device_count() : 0));

   /* LoadHMMSet */
   TIC;
   htkamd_mmf *mmf; CHECK(htkamd_mmf_create(&mmf));
   for (int i = 0; i < mmfs.n; i++) CHECK(htkamd_mmf_read(mmf, mmfs.v[i], NULL));
   CHECK(htkamd_mmf_finish(mmf, hmmList, hmmDir, hmmExt));
   const htkamd_model_desc *d = htkamd_mmf_desc(mmf);
   const int D = d->vecSize, H = d->numPhys;
   TOC(0);
   htkamd_model *model; CHECK(htkamd_model_create(d, &model));
   if (compat) CHECK(htkamd_model_set_compat(model, compat));
   TOC(1);
   int *shareMu = NULL, *shareVa = NULL;
   {  /* tied mean / variance vectors (~u ~v macros) */
      int *ms = (int *)malloc(sizeof(int) * (size_t)(d->numGauss + 1)), *vs = (int *)malloc(sizeof(int) * (size_t)(d->numGauss + 1));
      if (htkamd_mmf_sharing(mmf, ms, vs) > 0) { CHECK(htkamd_model_set_sharing(model, ms, vs)); shareMu = ms; shareVa = vs; }
      else { free(ms); free(vs); }
   }
   {  /* the update visits the models in the reference's scan order (it decides which user of a shared mean takes the mean-shift term) */
      const char **nm = (const char **)malloc(sizeof(char *) * (size_t)(d->numPhys + 1));
      int *order = (int *)malloc(sizeof(int) * (size_t)(d->numPhys + 1));
      for (int h = 0; h < d->numPhys; h++) nm[h] = htkamd_mmf_phys_name(mmf, h);
      CHECK(htkamd_hmm_scan_order(nm, d->numPhys, order));
      CHECK(htkamd_model_set_scan_order(model, order));
      free(nm); free(order);
   }
   htkamd_accs *accs; CHECK(htkamd_accs_create(model, &accs));
   htkamd_accs_layout lay; CHECK(htkamd_accs_get_layout(accs, &lay));
   double *vec = (double *)calloc(lay.total, sizeof(double));
   const char **physNames = (const char **)malloc(sizeof(char *) * (size_t)H);
   for (int h = 0; h < H; h++) physNames[h] = htkamd_mmf_phys_name(mmf, h);
   const char *tk = cfg_get(&cfg, "TARGETKIND");
   const int targetKind = kind_parse(tk ? tk : htkamd_mmf_parm_kind(mmf));
   if (trace & 1) {
      printf("HERest  %s Updating: %s%s%s%s\n\n", (uFlags & HTKAMD_UPMAP) ? "MAP" : "ML", (uFlags & HTKAMD_UPTRANS) ? "Transitions " : "", (uFlags & HTKAMD_UPMEANS) ? "Means " : "",
             (uFlags & HTKAMD_UPVARS) ? "Variances " : "", (uFlags & HTKAMD_UPMIXES) ? "MixWeights " : "");
      printf("%d Logical/%d Physical Models Loaded, VecSize=%d\n", htkamd_mmf_num_logical(mmf), H, D);
   }
   if (pruneInit < HTKAMD_NOPRUNE) {
      if (pruneInc != 0.0) printf("Pruning-On[%.1f %.1f %.1f]\n", pruneInit, pruneInc, pruneLim); else printf("Pruning-On[%.1f]\n", pruneInit);
   } else printf("Pruning-Off\n");

   if (parMode == 0) {
      /* the data arguments are accumulator files: LoadAccs adds every dump (HTrain.c:1625) */
      for (int i = 0; i < files.n; i++) CHECK(htkamd_accs_load_file_shared(d, vec, physNames, uFlags, shareMu, shareVa, files.v[i]));
      CHECK(htkamd_accs_upload_add(accs, vec, NULL));
   } else {
      htkamd_mlf *mlf = NULL;
      if (mlfPath) CHECK(htkamd_mlf_read(mlfPath, &mlf));
      htkamd_comm *comm = NULL;
      if (nRanks > 1) {
         /* Rendezvous through a file: [8-byte nonce of the run][128-byte RCCL id].  Rank 0 removes whatever a previous run left under
            the name before it writes (and again when it exits); the other ranks take a file only if it carries this run's nonce, so a
            stale id is waited out, not used.  Every wait is bounded: a rank that cannot meet the others within --rccl-timeout seconds
            (here, or in the all-reduce below when a rank has died on an error) exits non-zero instead of hanging. */
         unsigned char id[128];
         if (rank == 0) {
            unlink(rcclIdFile);
            CHECK(htkamd_comm_unique_id(id));
            char tmp[1024]; snprintf(tmp, sizeof(tmp), "%s.tmp", rcclIdFile);
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(&rcclNonce, 1, 8, f) != 8 || fwrite(id, 1, 128, f) != 128) DIE("cannot write %s", tmp);
            fclose(f); rename(tmp, rcclIdFile);
            g_rcclIdFile = rcclIdFile; atexit(remove_id_file);
         } else {
            int got = 0;
            for (int tries = 0; tries < rcclTimeout * 100 && !got; tries++) {
               FILE *f = fopen(rcclIdFile, "rb");
               unsigned long long n = 0;
               if (f) { got = fread(&n, 1, 8, f) == 8 && n == rcclNonce && fread(id, 1, 128, f) == 128; fclose(f); }
               if (!got) usleep(10000);
            }
            if (!got) DIE("rank %d: no RCCL id of this run (nonce %llu) in %s after %d s", rank, rcclNonce, rcclIdFile, rcclTimeout);
         }
         g_rank = rank; signal(SIGALRM, rendezvous_timeout); alarm((unsigned)rcclTimeout);
         CHECK(htkamd_comm_init(&comm, nRanks, rank, id));
         alarm(0);
      }
      /* this rank's shard: files rank, rank + R, ... (HERest -p semantics with the script file split round-robin) */
      strlist mine = {0};
      for (int i = rank; i < files.n; i += nRanks) sl_add(&mine, files.v[i]);
      htkamd_fb_config fc; memset(&fc, 0, sizeof(fc));
      fc.pruneInit = pruneInit; fc.pruneInc = pruneInc; fc.pruneLim = pruneLim; fc.minFrwdP = minFrwdP; fc.uFlags = uFlags; fc.scoreMode = scoreMode;
      /* The shard in batches.  With --iterations K > 1 everything a batch needs stays where the first iteration put it -- features in HBM,
         transcriptions as model indices, the batch tables of CreateInsts / SetBeamTaper in their context -- and the model never leaves
         the device between iterations: iteration 2.. cost the pass and the update, not the files. */
      typedef struct { obs_batch ob; int *labOff, *labs; int count, first, prepared; htkamd_fb *fb; } dev_batch;
      const int nBatch = (mine.n + batchN - 1) / batchN;
      dev_batch *bt = (dev_batch *)calloc((size_t)(nBatch ? nBatch : 1), sizeof(dev_batch));
      int settled = 0;                                 /* the ranks have agreed on the bf16 fallback */
      for (int it = 1; it <= nIter; it++) {
         int again;
         const double itStart = now_s();
         /* bound of a wait in a collective: --rccl-timeout on top of twice what this rank's own pass has taken so far (the shards are even;
            a rank repeating its iteration as bf16 takes twice as long): a healthy job with long passes is not mistaken for a dead rank */
#define COLL_ALARM() alarm((unsigned)rcclTimeout + (unsigned)(2.0 * (now_s() - itStart)) + 1u)
         do {                                          /* twice only when the fp16 scoring path reports data outside its range */
         again = 0;
         CHECK(htkamd_accs_zero(accs, NULL));
         /* Batches stay resident only across --iterations: a single pass streams them in bounded memory in every scoring mode.  Should
            the fp16 range check trip (rare), the iteration starts over as bf16 and the batches already released are read again. */
         const int keep = nIter > 1;
         for (int bi = 0; bi <= nBatch && !again; bi++) {
            if (bi < nBatch) {
               dev_batch *B = &bt[bi];
               tic_ = now_s();
               if (!B->fb) {
                  B->first = bi * batchN; B->count = (mine.n - B->first < batchN) ? mine.n - B->first : batchN;
                  load_observations(&mine, B->first, B->count, targetKind, &cfg, &B->ob);
                  if (B->ob.cols != D) DIE("observations have %d components, the models %d", B->ob.cols, D);
                  int capLab = 0;
                  B->labOff = (int *)calloc((size_t)B->count + 1, sizeof(int));
                  for (int u = 0; u < B->count; u++) {
                     char lab[2048];
                     make_fn(mine.v[B->first + u], labDir, labExt, lab, sizeof(lab));
                     htkamd_labels *L = NULL; const htkamd_labels *Lc = NULL;
                     if (mlf) { Lc = htkamd_mlf_find(mlf, lab); if (!Lc) DIE("%s: no entry in the master label file %s", lab, mlfPath); }
                     else { CHECK(htkamd_labels_read(lab, &L)); Lc = L; }
                     const int n = htkamd_labels_count(Lc);
                     if (n == 0) fprintf(stderr, "WARNING [-7325] LoadUtterance: No labels in file %s\n", lab);
                     if (B->labOff[u] + n > capLab) { capLab = (B->labOff[u] + n) * 2 + 64; B->labs = (int *)realloc(B->labs, sizeof(int) * (size_t)capLab); }
                     for (int i = 0; i < n; i++) {
                        const int h = htkamd_mmf_find_logical(mmf, htkamd_labels_name(Lc, i));
                        if (h < 0) DIE("[7321] CreateInsts: Unknown label %s in %s", htkamd_labels_name(Lc, i), lab);
                        B->labs[B->labOff[u] + i] = h;
                     }
                     B->labOff[u + 1] = B->labOff[u] + n;
                     if (L) htkamd_labels_free(L);
                  }
                  CHECK(htkamd_fb_create(model, &B->fb));
               }
               TOC(2);
               if (!B->prepared || !htkamd_fb_prepared_current(B->fb)) {      /* the update changed a minimum duration: tables again */
                  B->prepared = 1;
                  htkamd_batch_desc b = {B->count, B->ob.dX, B->ob.frameOff, B->labOff, B->labs};
                  CHECK(htkamd_fb_prepare(B->fb, &b, NULL));
               }
               TOC(3);
               CHECK(htkamd_fb_execute(B->fb, &fc, accs, NULL));            /* asynchronous: the next batch's files are read while this one runs */
            }
            if (bi > 0) {                                                     /* results of the batch before */
               dev_batch *B = &bt[bi - 1];
               double *pr = (double *)malloc(sizeof(double) * (size_t)B->count); int *st = (int *)malloc(sizeof(int) * (size_t)B->count);
               tic_ = now_s();
               const int rcr = htkamd_fb_results(B->fb, pr, st, NULL);
               if (rcr == HTKAMD_ERANGE && (fc.scoreMode & HTKAMD_SCORE_F16)) {
                  /* a value the fp16 x 2 scores cannot hold: the whole iteration again on the bf16 x 3 path (same tolerance class, fp32's range) */
                  fprintf(stderr, "WARNING: %s\n  herest: repeating the iteration with the bf16 x 3 scoring path\n", htkamd_last_error());
                  fc.scoreMode = (fc.scoreMode & ~HTKAMD_SCORE_F16) | HTKAMD_SCORE_BF16;
                  CHECK(htkamd_stream_sync(NULL));
                  free(pr); free(st);
                  again = 1;
                  break;
               }
               CHECK(rcr);
               TOC(4);
               for (int u = 0; u < B->count; u++) {
                  if (trace & 1) printf(" Processing Data: %s\n", mine.v[B->first + u]);
                  if (st[u] == HTKAMD_UTT_OK) { if (trace & 1) printf(" Utterance prob per frame = %e\n", pr[u] / (B->ob.frameOff[u + 1] - B->ob.frameOff[u])); }
                  else if (st[u] == HTKAMD_UTT_SKIPPED) fprintf(stderr, "WARNING [-7324] StepBack: File %s - bad data or over pruning\n", mine.v[B->first + u]);
                  else DIE("[%d] forward-backward failed on %s", -st[u], mine.v[B->first + u]);
               }
               free(pr); free(st);
               if (!keep) { free(B->labOff); free(B->labs); free_observations(&B->ob); htkamd_fb_destroy(B->fb); B->fb = NULL; B->labOff = NULL; B->labs = NULL; B->prepared = 0; }
            }
         }
         if (comm && !again && (scoreMode & HTKAMD_SCORE_F16) && !settled) {
            /* the ranks take the fp16 -> bf16 decision alike: no sum of statistics of two arithmetics.  One call per iteration on every
               rank until a fallback is agreed (a rank that fell back repeats its iteration FIRST and says so here; the others repeat
               theirs after this call, without a second one), none afterwards: the counts of collectives stay equal. */
            int fell = !(fc.scoreMode & HTKAMD_SCORE_F16);
            COLL_ALARM();
            CHECK(htkamd_comm_agree_max(comm, &fell, NULL));
            alarm(0);
            if (fell) {
               settled = 1;
               if (fc.scoreMode & HTKAMD_SCORE_F16) {
                  fprintf(stderr, "WARNING: herest rank %d: another rank's data does not fit the fp16 scores' range: repeating the iteration with the bf16 x 3 scoring path\n", rank);
                  fc.scoreMode = (fc.scoreMode & ~HTKAMD_SCORE_F16) | HTKAMD_SCORE_BF16;
                  again = 1;
               }
            }
         }
         } while (again);
         if (comm) {
            COLL_ALARM();                                   /* a rank that died before this point would leave the others in the collective for ever */
            CHECK(htkamd_accs_allreduce_wire(accs, comm, wire, NULL)); CHECK(htkamd_stream_sync(NULL));
            alarm(0);
         }
         if (it < nIter) {
            /* an intermediate iteration: UpdateModels where the model is, HERest's summary lines, nothing written */
            tic_ = now_s();
            CHECK(htkamd_accs_download(accs, vec, NULL));
            htkamd_update_config uc; memset(&uc, 0, sizeof(uc));
            uc.minEgs = minEgs; uc.minVar = minVar; uc.mixWeightFloor = mixFloor; uc.uFlags = uFlags; uc.varFloor = htkamd_mmf_var_floor(mmf); uc.singleProcess = (parMode == -1);   /* every iteration as a HERest process of its own would make it */
            htkamd_update_stats us;
            if (uFlags & HTKAMD_UPMAP) DIE("herest: --iterations with -u p (MAP) is not supported: one HERest pass per prior");
            CHECK(htkamd_model_update_device(model, accs, &uc, &us, NULL));
            TOC(5);
            if (rank == 0) {
               if (us.nFloorVar > 0) printf("Total %d floored variance elements in %d different mixes\n", us.nFloorVar, us.nFloorVarMix);
               printf("Iteration %d of %d complete - average log prob per frame = %e\n", it, nIter, vec[lay.totalPr] / vec[lay.totalT]);
            }
         }
      }
      for (int bi = 0; bi < nBatch; bi++) if (bt[bi].fb) { free(bt[bi].labOff); free(bt[bi].labs); free_observations(&bt[bi].ob); htkamd_fb_destroy(bt[bi].fb); }
      free(bt);
      if (comm) htkamd_comm_destroy(comm);
      CHECK(htkamd_accs_download(accs, vec, NULL));
      if (mlf) htkamd_mlf_free(mlf);
   }

   if (parMode > 0) {
      char fn[2048];
      snprintf(fn, sizeof(fn), "%s/HER%d.acc", outDir ? outDir : ".", parMode);
      CHECK(htkamd_accs_dump_file_shared(d, vec, physNames, uFlags, shareMu, shareVa, fn));
      if (trace & 1) printf("Accumulators dumped to %s\n", fn);
      return 0;
   }
   if (statsFile) CHECK(htkamd_stats_write_file(d, vec, physNames, statsFile));

   /* UpdateModels on the device, then SaveHMMSet (every rank computes the same models; rank 0 writes them) */
   tic_ = now_s();
   htkamd_update_config uc; memset(&uc, 0, sizeof(uc));
   uc.minEgs = minEgs; uc.minVar = minVar; uc.mixWeightFloor = mixFloor; uc.uFlags = uFlags; uc.varFloor = htkamd_mmf_var_floor(mmf);
   uc.singleProcess = (parMode == -1);
   htkamd_update_stats us;
   if (uFlags & HTKAMD_UPMAP) {
      /* MAPUpdateModels reads HMap's own configuration, not HERest's switches (InitMap HMap.c:88-104) */
      const char *v;
      uc.minEgs = (v = cfg_get_mod(&cfg, "HMAP", "MINEGS")) ? atoi(v) : 0;
      uc.minVar = (v = cfg_get_mod(&cfg, "HMAP", "MINVAR")) ? (float)atof(v) : 0.0f;
      uc.mixWeightFloor = (v = cfg_get_mod(&cfg, "HMAP", "MIXWEIGHTFLOOR")) ? (float)(1.0e-5 * atof(v)) : 0.0f;
      uc.mapTau = (v = cfg_get_mod(&cfg, "HMAP", "MAPTAU")) ? (float)atof(v) : 20.0f;
      uc.mapMinObs = (v = cfg_get_mod(&cfg, "HMAP", "MINOBS")) ? (float)atof(v) : 0.0f;
      CHECK(htkamd_model_update(model, accs, vec, &uc, &us));
      if ((v = cfg_get_mod(&cfg, "HMAP", "TRACE")) && (atoi(v) & 1)) {
         int totM = 0;
         { int *ms = (int *)malloc(sizeof(int) * (size_t)(d->numGauss + 1)), *vs = (int *)malloc(sizeof(int) * (size_t)(d->numGauss + 1));
           const int shared = htkamd_mmf_sharing(mmf, ms, vs);
           for (int g = 0; g < d->numGauss; g++) {                                             /* TotMixInSet: distinct mean vectors */
              int seen = 0;
              if (shared > 0 && ms[g] >= 0) for (int k = 0; k < g; k++) if (ms[k] == ms[g]) { seen = 1; break; }   /* ms: number of the ~u macro, -1 = private */
              if (!seen) totM++;
           }
           free(ms); free(vs); }
         printf("Observed components (means) %d of %d: %.2f\n", us.nMapObserved, totM, 100 * (float)us.nMapObserved / (float)totM);
         if (us.nFloorVar > 0) printf("Total %d floored variance elements in %d different mixes\n", us.nFloorVar, us.nFloorVarMix);
      }
   } else CHECK(htkamd_model_update_device(model, accs, &uc, &us, NULL));      /* tied ~u ~v vectors included (pooled statistics, update.hip) */
   TOC(5);
   if (us.nSkippedHmm > 0) fprintf(stderr, "WARNING [-2331] UpdateModels: %d models had fewer than %d examples and were copied\n", us.nSkippedHmm, minEgs);
   if (rank == 0) {
      float *mean = (float *)malloc(sizeof(float) * (size_t)d->numGauss * D), *var = (float *)malloc(sizeof(float) * (size_t)d->numGauss * D);
      float *gc = (float *)malloc(sizeof(float) * (size_t)d->numGauss), *wt = (float *)malloc(sizeof(float) * (size_t)d->numComp);
      float *tp = (float *)malloc(sizeof(float) * (size_t)d->transOff[d->numTrans]);
      CHECK(htkamd_model_get_params(model, mean, var, gc, wt, tp));
      char one[2048]; const char *oneFile = NULL;
      if (mmfs.n > 0) { make_fn(mmfs.v[0], outDir ? outDir : ".", NULL, one, sizeof(one)); oneFile = one; }
      if (mmfs.n > 1) {
         /* several master files (-H macros -H hmmdefs): every macro back to the file it came from, as SaveHMMSet does, so that the
            next iteration's -H dir/macros -H dir/hmmdefs finds them */
         char **outs = (char **)malloc(sizeof(char *) * (size_t)mmfs.n);
         for (int k = 0; k < mmfs.n; k++) { outs[k] = (char *)malloc(2048); make_fn(mmfs.v[k], outDir ? outDir : ".", NULL, outs[k], 2048); }
         CHECK(htkamd_mmf_write_sources(mmf, mean, var, gc, wt, tp, (const char *const *)outs, mmfs.n, outDir ? outDir : ".", binary));
         for (int k = 0; k < mmfs.n; k++) free(outs[k]);
         free(outs);
      }
      else if (binary) CHECK(htkamd_mmf_write_binary(mmf, mean, var, gc, wt, tp, oneFile, oneFile ? NULL : (outDir ? outDir : ".")));
      else CHECK(htkamd_mmf_write(mmf, mean, var, gc, wt, tp, oneFile, oneFile ? NULL : (outDir ? outDir : ".")));
      TOC(6);
      if (us.nFloorVar > 0 && !(uFlags & HTKAMD_UPMAP)) printf("Total %d floored variance elements in %d different mixes\n", us.nFloorVar, us.nFloorVarMix);
      if (trace & 1) printf("Saving hmm's to %s %s\n", oneFile ? "MMF" : "dir", oneFile ? oneFile : (outDir ? outDir : "Current"));
      printf("Reestimation complete - average log prob per frame = %e\n", vec[lay.totalPr] / vec[lay.totalT]);
      printf("     - total frames seen          = %e\n", vec[lay.totalT]);
   }
   if (trace & 040000) for (int k = 0; k < 7; k++) printf("Timing: %-22s %8.3f s\n", g_tn[k], g_t[k]);
   htkamd_accs_destroy(accs); htkamd_model_destroy(model); htkamd_mmf_destroy(mmf);
   return 0;
}
