/* hvite.c -- Viterbi recognition and forced alignment on the MI355X: the command-line program a user of the reference's HVite switches to.
 *
 *   hvite [options] -w net.slf dict hmmList dataFiles...          recognition over a word network
 *   hvite [options] -a [-b sil] dict hmmList dataFiles...         alignment of each file's word-level transcription
 *
 * Options (the subset of HTKBook ref.tex "HVite" that SURVEY.md 8(b) lists; same letters, meaning and defaults, HVite.c:227-420):
 *   -C cf        configuration file (TARGETKIND, DELTAWINDOW, ACCWINDOW, THIRDWINDOW, V1COMPAT, SIMPLEDIFFS)
 *   -S scp       script file with further data files            -H mmf / -d dir / -x ext   model sources
 *   -w net       recognition network (SLF word lattice)         -a            align against the label files instead
 *   -b word      alignment: boundary word at both ends          -L dir / -X ext / -I mlf   transcriptions to align (default .lab)
 *   -i mlf       write one master label file                    -l dir / -y ext            or one label file per input (default .rec)
 *   -m           model-level labels                             -f            state-level labels (implies model level as auxiliary)
 *   -o chars     output format S W T N X C M (HVite -o)
 *   -t f         general beam (0 = off)                         -v f          word-end beam           -u N   maximum active models (0 = off)
 *   -s f         grammar scale (1.0)    -p f   word insertion penalty (0.0)    -r f   pronunciation scale (1.0)
 *   -T N         trace (1: one line per file)
 * Beyond the reference:  --score exact|fast|fastest (default exact: paths and scores are the reference's bit for bit),  --batch N.
 * Every file's labels are "start end name score [aux ...]" in 100 ns units as TranscriptionFromLattice / FormatTranscription
 * (HRec.c:2176,2368) produce them; a file in which no token survives gets no entry and a "No tokens survived" line on stderr.
 */
#include "cli_common.h"

static int out_flags(const char *s)
{
   int f = 0;
   for (; *s; s++)
      switch (*s) {
      case 'S': f |= HTKAMD_OUT_NOSCORES; break; case 'W': f |= HTKAMD_OUT_NOWORDS; break; case 'T': f |= HTKAMD_OUT_NOTIMES; break;
      case 'N': f |= HTKAMD_OUT_NORMSCORES; break; case 'X': f |= HTKAMD_OUT_TRISTRIP; break; case 'C': f |= HTKAMD_OUT_CENTRE; break;
      case 'M': f |= HTKAMD_OUT_NOMODELS; break;
      default: DIE("-o: unknown format character %c (S W T N X C M)", *s);
      }
   return f;
}

/* labels of one decoded utterance -> transcription, at word / model / state level */
static void emit(htkamd_trans *tr, const htkamd_net *net, const htkamd_mmf *mmf, htkamd_viterbi *vit, const float *dX, int D, int frame0, int T,
                 int nW, const int *wPron, const int *wStart, const int *wEnd, const float *wScore, const float *wLm,
                 double period, int models, int states, float lmScale, float wordPen, float alignBeam)
{
   if (!models && !states) {
      for (int w = 0; w < nW; w++) {
         const char *sym = htkamd_net_out_sym(net, wPron[w]);
         if (!sym || !sym[0]) continue;                        /* words without output symbol leave no label (HRec.c:2342-2356) */
         CHECK(htkamd_trans_add(tr, wStart[w] * period, wEnd[w] * period, sym, wScore[w], NULL, 0, NULL, 0));
      }
      return;
   }
   /* the model chain of the recognised pronunciations, aligned against the same frames: for a fixed word sequence the LM terms are
      constants, so the best alignment of the chain is the decoder's path (include/htk_amd.h, htkamd_decoder_run) */
   int nQ = 0, cap = 0, *chain = NULL, *wordOfQ = NULL;
   for (int w = 0; w < nW; w++) {
      int tmp[256], prev = -1, next = -1;                      /* neighbours with phones: cross-word networks name a word's edge models by them */
      for (int z = w - 1; z >= 0 && prev < 0; z--) if (htkamd_net_pron_models(net, wPron[z], tmp, 0) > 0) prev = wPron[z];
      for (int z = w + 1; z < nW && next < 0; z++) if (htkamd_net_pron_models(net, wPron[z], tmp, 0) > 0) next = wPron[z];
      const int n = htkamd_net_seq_models(net, wPron[w], prev, next, tmp, 256);
      if (n < 0) DIE("%s", htkamd_last_error());
      if (n > 256) DIE("pronunciation with %d models", n);
      if (nQ + n > cap) { cap = (nQ + n) * 2 + 64; chain = (int *)realloc(chain, sizeof(int) * (size_t)cap); wordOfQ = (int *)realloc(wordOfQ, sizeof(int) * (size_t)cap); }
      for (int k = 0; k < n; k++) { chain[nQ] = tmp[k]; wordOfQ[nQ] = (k == 0) ? w : -1; nQ++; }
   }
   if (nQ == 0) { free(chain); free(wordOfQ); return; }
   int frameOff[2] = {0, T}, labOff[2] = {0, nQ};
   htkamd_batch_desc b = {1, dX + (size_t)frame0 * D, frameOff, labOff, chain};
   CHECK(htkamd_viterbi_align(vit, &b, alignBeam, NULL));
   size_t nSeg, nMod; CHECK(htkamd_viterbi_sizes(vit, &nSeg, &nMod));
   int *segS = (int *)malloc(sizeof(int) * nSeg), *segE = (int *)malloc(sizeof(int) * nSeg), *modS = (int *)malloc(sizeof(int) * nMod), *modE = (int *)malloc(sizeof(int) * nMod), st;
   double *segSc = (double *)malloc(sizeof(double) * nSeg), *modSc = (double *)malloc(sizeof(double) * nMod), tot;
   CHECK(htkamd_viterbi_results(vit, segS, segE, segSc, modS, modE, modSc, &tot, &st, NULL));
   const htkamd_model_desc *d = htkamd_mmf_desc(mmf);
   size_t k = 0;
   for (int q = 0; q < nQ; q++) {
      const int h = chain[q], ns = d->hmmStateOff[h + 1] - d->hmmStateOff[h], w = wordOfQ[q];
      const char *mname = htkamd_mmf_phys_name(mmf, h);
      const char *wname = (w >= 0) ? htkamd_net_word_name(net, wPron[w]) : NULL;     /* -m / -f label the first model of a word with the word's NAME */
      /* LArcTotLMLike (HNet.h:252): lmlike*lmscale + wdpenalty, in float then double as the reference evaluates it */
      const float waux = (w >= 0) ? (float)((double)(float)(wLm[w] * lmScale) + (double)wordPen) : 0.0f;
      if (!states) CHECK(htkamd_trans_add(tr, modS[q] * period, modE[q] * period, mname, (float)modSc[q], wname, wname ? waux : 0.0f, NULL, 0));
      int first = 1;
      for (int j = 0; j < ns; j++, k++) {
         if (!states || segS[k] < 0) continue;
         char sname[600];
         if (models) {
            snprintf(sname, sizeof(sname), "s%d", j + 2);
            CHECK(htkamd_trans_add(tr, segS[k] * period, segE[k] * period, sname, (float)segSc[k], first ? mname : NULL, first ? (float)modSc[q] : 0.0f,
                                   first ? wname : NULL, (first && wname) ? waux : 0.0f));
         } else {
            snprintf(sname, sizeof(sname), "%s[%d]", mname, j + 2);
            CHECK(htkamd_trans_add(tr, segS[k] * period, segE[k] * period, sname, (float)segSc[k], first ? wname : NULL, (first && wname) ? waux : 0.0f, NULL, 0));
         }
         first = 0;
      }
   }
   free(segS); free(segE); free(modS); free(modE); free(segSc); free(modSc); free(chain); free(wordOfQ);
}

/* HShell's argument classes (NextArg, HShell.c:1158-1200): an argument is a number when the whole of it parses as one; *isInt: no
   fraction, no exponent */
static int is_number(const char *s, int *isInt)
{
   const char *p = s;
   int digits = 0, frac = 0;
   if (*p == '+' || *p == '-') p++;
   while (isdigit((unsigned char)*p)) { p++; digits++; }
   if (*p == '.') { frac = 1; p++; while (isdigit((unsigned char)*p)) { p++; digits++; } }
   if (!digits) return 0;
   if (*p == 'e' || *p == 'E') {
      const char *q = p + 1;
      int ed = 0;
      if (*q == '+' || *q == '-') q++;
      while (isdigit((unsigned char)*q)) { q++; ed++; }
      if (!ed) return 0;
      frac = 1; p = q;
   }
   if (*p) return 0;
   if (isInt) *isInt = !frac;
   return 1;
}

int main(int argc, char **argv)
{
   args a = {argc, 1, argv};
   config cfg; memset(&cfg, 0, sizeof(cfg));
   strlist mmfs = {0}, files = {0};
   const char *hmmDir = NULL, *hmmExt = NULL, *netPath = NULL, *labDir = NULL, *labExt = "lab", *mlfIn = NULL, *mlfOut = NULL, *outDir = NULL, *outExt = "rec", *boundary = NULL;
   float genBeam = 0.0f, wordBeam = 0.0f, lmScale = 1.0f, wordPen = 0.0f, prScale = 1.0f;
   float genBeamInc = 0.0f, genBeamLim = 0.0f, tmBeam = 10.0f;             /* -t f [i l]: alignment retries a file with wider beams (HVite.c:308-322, 900-913) */
   int align = 0, models = 0, states = 0, oflags = 0, trace = 0, scoreMode = HTKAMD_SCORE_EXACT, batchN = 1024, maxActive = 0;
   int nToks = 0, nTrans = 1, latFmt = 0;
   const char *latExt = NULL;
   const char *sw;

   while (a.at < a.argc && is_switch(a.argv[a.at])) {
      if (!strncmp(a.argv[a.at], "--", 2)) {
         const char *lo = a.argv[a.at++] + 2;
         if (!strcmp(lo, "score")) {
            const char *m = str_arg(&a, "-score");
            scoreMode = !strcmp(m, "exact") ? HTKAMD_SCORE_EXACT : !strcmp(m, "fast") ? HTKAMD_SCORE_MFMA : !strcmp(m, "fastest") ? HTKAMD_SCORE_BF16 : -1;
            if (scoreMode < 0) DIE("--score: exact | fast | fastest");
         } else if (!strcmp(lo, "batch")) batchN = atoi(str_arg(&a, "-batch"));
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
      case 'w': netPath = str_arg(&a, sw); break;
      case 'a': align = 1; break;
      case 'b': boundary = str_arg(&a, sw); break;
      case 'L': labDir = str_arg(&a, sw); break;
      case 'X': labExt = str_arg(&a, sw); break;
      case 'I': mlfIn = str_arg(&a, sw); break;
      case 'i': mlfOut = str_arg(&a, sw); break;
      case 'l': outDir = str_arg(&a, sw); break;
      case 'y': outExt = str_arg(&a, sw); break;
      case 'm': models = 1; break;
      case 'f': states = 1; break;
      case 'o': oflags = out_flags(str_arg(&a, sw)); break;
      case 't':                                             /* -t f [i l] (HVite.c:308-322): the two further values only when the next argument IS a number */
         genBeam = (float)flt_arg(&a, sw);
         genBeamInc = 0.0f; genBeamLim = genBeam;
         if (a.at < a.argc && is_number(a.argv[a.at], NULL)) {
            genBeamInc = (float)flt_arg(&a, sw);
            if (!(a.at < a.argc && is_number(a.argv[a.at], NULL))) DIE("hvite -t: f [i l] -- the limit is missing");
            genBeamLim = (float)flt_arg(&a, sw);
            if (genBeamLim < genBeam + genBeamInc) { genBeamLim = genBeam; genBeamInc = 0.0f; }
         }
         break;
      case 'c': tmBeam = (float)flt_arg(&a, sw); break;     /* tied mixture pruning threshold (HVite.c:255) */
      case 'v': wordBeam = (float)flt_arg(&a, sw); break;
      case 's': lmScale = (float)flt_arg(&a, sw); break;
      case 'p': wordPen = (float)flt_arg(&a, sw); break;
      case 'r': prScale = (float)flt_arg(&a, sw); break;
      case 'u': maxActive = atoi(str_arg(&a, sw)); break;
      case 'n':                                              /* -n i [N]: i tokens per state, N-best transcriptions (HVite.c:287-291) */
         nToks = atoi(str_arg(&a, sw));
         { int isInt = 0;                                   /* NextArg() == INTARG (HVite.c:289): the WHOLE argument is an integer */
           if (a.at < a.argc && is_number(a.argv[a.at], &isInt) && isInt) nTrans = atoi(a.argv[a.at++]); }
         break;
      case 'z': latExt = str_arg(&a, sw); break;
      case 'q':
         for (const char *c = str_arg(&a, sw); *c; c++)
            switch (*c) {
            case 't': latFmt |= HTKAMD_LAT_TIMES; break; case 'v': latFmt |= HTKAMD_LAT_PRON; break; case 'a': latFmt |= HTKAMD_LAT_ACLIKE; break;
            case 'l': latFmt |= HTKAMD_LAT_LMLIKE; break; case 'r': latFmt |= HTKAMD_LAT_PRLIKE; break;
            default: DIE("hvite -q: only t v a l r are supported");
            }
         break;
      case 'T': trace = atoi(str_arg(&a, sw)); break;
      default: DIE("hvite: unknown switch -%s", sw);
      }
   }
   if (a.at + 2 > a.argc) DIE("hvite: dictionary and HMM list expected");
   const char *dictPath = a.argv[a.at++], *hmmList = a.argv[a.at++];
   while (a.at < a.argc) sl_add(&files, a.argv[a.at++]);
   if (files.n == 0) DIE("hvite: no data files");
   if (!align && !netPath) DIE("hvite: either -w net or -a");
   if (nToks == 1 || nToks > 8) DIE("hvite -n: 2..8 tokens per state");
   if (nToks > 1 && align) DIE("hvite: alignment using multiple tokens is not supported");       /* HVite.c:448 (-m / -f with -n: as the reference built with -DPHNALG, alignment records inside the lattice arcs) */
   if (latExt && nToks < 2) DIE("hvite -z: lattices need -n i with i > 1");
   if (htkamd_device_count() <= 0) DIE("hvite: no HIP device (the MI355X path has no CPU fallback)");

   htkamd_mmf *mmf; CHECK(htkamd_mmf_create(&mmf));
   for (int i = 0; i < mmfs.n; i++) CHECK(htkamd_mmf_read(mmf, mmfs.v[i], NULL));
   CHECK(htkamd_mmf_finish(mmf, hmmList, hmmDir, hmmExt));
   const htkamd_model_desc *d = htkamd_mmf_desc(mmf);
   const int D = d->vecSize;
   htkamd_model *model; CHECK(htkamd_model_create(d, &model));
   if (d->hsKind == HTKAMD_HS_TIED) CHECK(htkamd_model_set_tm_beam(model, tmBeam));
   htkamd_viterbi *vit; CHECK(htkamd_viterbi_create(model, &vit));
   const char *tk = cfg_get(&cfg, "TARGETKIND");
   const int targetKind = kind_parse(tk ? tk : htkamd_mmf_parm_kind(mmf));
   htkamd_net *net = NULL; htkamd_decoder *dec = NULL;
   const int netFlags = (cfg_bool(&cfg, "ALLOWXWRDEXP", 0) ? HTKAMD_NET_ALLOWXWRDEXP : 0) | (cfg_bool(&cfg, "FORCECXTEXP", 0) ? HTKAMD_NET_FORCECXTEXP : 0) |
                        (cfg_bool(&cfg, "FORCELEFTBI", 0) ? HTKAMD_NET_FORCELEFTBI : 0) | (cfg_bool(&cfg, "FORCERIGHTBI", 0) ? HTKAMD_NET_FORCERIGHTBI : 0);     /* HNet.c:122-127 */
   if (!align) { CHECK(htkamd_net_build_ex(netPath, dictPath, mmf, netFlags, &net)); CHECK(htkamd_decoder_create(model, htkamd_net_get(net), lmScale, &dec)); }
   htkamd_mlf *mlf = NULL; if (mlfIn) CHECK(htkamd_mlf_read(mlfIn, &mlf));
   htkamd_mlf_out *mout = NULL; if (mlfOut) CHECK(htkamd_mlf_out_open(mlfOut, &mout));
   htkamd_decode_config dc; memset(&dc, 0, sizeof(dc));
   dc.genBeam = genBeam > 0 ? genBeam : 1.0e10f; dc.wordBeam = wordBeam > 0 ? wordBeam : 1.0e10f;
   dc.lmScale = lmScale; dc.wordPen = wordPen; dc.prScale = prScale; dc.scoreMode = scoreMode; dc.maxActive = maxActive;
   const int maxWords = 4096;

   for (int first = 0; first < files.n; first += (align ? 1 : batchN)) {
      const int count = align ? 1 : ((files.n - first < batchN) ? files.n - first : batchN);
      obs_batch ob; memset(&ob, 0, sizeof(ob));
      load_observations(&files, first, count, targetKind, &cfg, &ob);
      if (ob.cols != D) DIE("observations have %d components, the models %d", ob.cols, D);
      htkamd_net *unet = net; htkamd_decoder *udec = dec;
      if (align) {                                              /* DoAlignment (HVite.c:830): the network of this file's transcription */
         char lab[2048];
         make_fn(files.v[first], labDir, labExt, lab, sizeof(lab));
         htkamd_labels *L = NULL; const htkamd_labels *Lc = NULL;
         if (mlf) { Lc = htkamd_mlf_find(mlf, lab); if (!Lc) DIE("%s: no entry in the master label file %s", lab, mlfIn); }
         else { CHECK(htkamd_labels_read(lab, &L)); Lc = L; }
         const int n = htkamd_labels_count(Lc);
         const char **words = (const char **)malloc(sizeof(char *) * (size_t)(n ? n : 1));
         for (int i = 0; i < n; i++) words[i] = htkamd_labels_name(Lc, i);
         CHECK(htkamd_net_build_words(words, n, boundary, dictPath, mmf, &unet));
         CHECK(htkamd_decoder_create(model, htkamd_net_get(unet), lmScale, &udec));
         free(words);
         if (L) htkamd_labels_free(L);
      }
      if (nToks > 1) {
         /* ---- N-best: token sets on the device, lattice per file, then WriteLattice / the N most likely transcriptions on the host */
         const int maxN = 65536, maxA = 262144;
         const int per = (count < 32) ? count : 32;                    /* lattices of 32 files at a time */
         for (int b0 = 0; b0 < count; b0 += per) {
            const int nb = (count - b0 < per) ? count - b0 : per;
            int *nN = (int *)calloc((size_t)nb, sizeof(int)), *nA = (int *)calloc((size_t)nb, sizeof(int));
            int *nodeFrame = (int *)malloc(sizeof(int) * (size_t)nb * maxN), *nodePron = (int *)malloc(sizeof(int) * (size_t)nb * maxN);
            double *nodeLike = (double *)malloc(sizeof(double) * (size_t)nb * maxN);
            int *aS = (int *)malloc(sizeof(int) * (size_t)nb * maxA), *aE = (int *)malloc(sizeof(int) * (size_t)nb * maxA);
            float *aAc = (float *)malloc(sizeof(float) * (size_t)nb * maxA), *aLm = (float *)malloc(sizeof(float) * (size_t)nb * maxA), *aPr = (float *)malloc(sizeof(float) * (size_t)nb * maxA);
            htkamd_lattice_out lo; memset(&lo, 0, sizeof(lo));
            lo.nNodes = nN; lo.nArcs = nA; lo.nodeFrame = nodeFrame; lo.nodePron = nodePron; lo.nodeLike = nodeLike; lo.arcStart = aS; lo.arcEnd = aE; lo.arcAc = aAc; lo.arcLm = aLm; lo.arcPr = aPr;
            int *fo = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
            for (int u = 0; u <= nb; u++) fo[u] = ob.frameOff[b0 + u] - ob.frameOff[b0];
            /* -m / -f together with -n: alignment records inside the arcs (LatFromPaths' lAlign) */
            const int alMode = (models ? 1 : 0) | (states ? 2 : 0), maxAl = alMode ? 4 * maxA : 0;
            htkamd_lattice_align_out ao; memset(&ao, 0, sizeof(ao));
            if (alMode) {
               ao.arcAlignOff = (int *)malloc(sizeof(int) * (size_t)nb * (maxA + 1)); ao.alState = (int *)malloc(sizeof(int) * (size_t)nb * maxAl);
               ao.alModel = (int *)malloc(sizeof(int) * (size_t)nb * maxAl); ao.alDur = (int *)malloc(sizeof(int) * (size_t)nb * maxAl); ao.alLike = (float *)malloc(sizeof(float) * (size_t)nb * maxAl);
               CHECK(htkamd_decoder_run_lattice_align(dec, &dc, nToks, dc.genBeam, alMode, ob.dX + (size_t)ob.frameOff[b0] * ob.cols, fo, nb, maxN, maxA, maxAl, &lo, &ao, NULL));
            } else
            CHECK(htkamd_decoder_run_lattice(dec, &dc, nToks, dc.genBeam, ob.dX + (size_t)ob.frameOff[b0] * ob.cols, fo, nb, maxN, maxA, &lo, NULL));   /* nBeam = genBeam (HVite.c:546) */
            for (int u = 0; u < nb; u++) {
               const char *fn = files.v[first + b0 + u];
               if (nN[u] == -1) { fprintf(stderr, "No tokens survived to final node of network: %s\n", fn); continue; }
               if (nN[u] < 0) DIE("the lattice of %s does not fit (%d nodes / %d arcs of room)", fn, maxN, maxA);
               htkamd_lattice lat; memset(&lat, 0, sizeof(lat));
               lat.nNodes = nN[u]; lat.nArcs = nA[u]; lat.nodeFrame = nodeFrame + (size_t)u * maxN; lat.nodePron = nodePron + (size_t)u * maxN; lat.nodeLike = nodeLike + (size_t)u * maxN;
               lat.arcStart = aS + (size_t)u * maxA; lat.arcEnd = aE + (size_t)u * maxA; lat.arcAc = aAc + (size_t)u * maxA; lat.arcLm = aLm + (size_t)u * maxA; lat.arcPr = aPr + (size_t)u * maxA;
               lat.lmScale = lmScale; lat.wordPen = wordPen; lat.prScale = prScale; lat.frameDur = (double)ob.period * 1.0e-7;
               char out[2048];
               htkamd_lattice_align la; memset(&la, 0, sizeof(la));
               if (alMode) {
                  la.arcAlignOff = ao.arcAlignOff + (size_t)u * (maxA + 1); la.alState = ao.alState + (size_t)u * maxAl; la.alModel = ao.alModel + (size_t)u * maxAl;
                  la.alDur = ao.alDur + (size_t)u * maxAl; la.alLike = ao.alLike + (size_t)u * maxAl; la.models = models;
               }
               if (latExt) {
                  make_fn(fn, outDir, latExt, out, sizeof(out));
                  if (alMode) CHECK(htkamd_lattice_write_align(&lat, &la, mmf, net, out, fn, netPath, dictPath, latFmt ? latFmt : HTKAMD_LAT_DEFAULT_ALIGN));
                  else CHECK(htkamd_lattice_write(&lat, net, out, fn, netPath, dictPath, latFmt ? latFmt : HTKAMD_LAT_DEFAULT));
               }
               /* "only output 1-best transcription if generating lattices" (HVite.c:797) */
               const int want = (nTrans > 1 && latExt) ? 1 : nTrans;
               int nAlt = 0, *altLen = (int *)malloc(sizeof(int) * (size_t)want), *altArcs = (int *)malloc(sizeof(int) * (size_t)want * maxWords);
               CHECK(htkamd_lattice_nbest(&lat, net, want, maxWords, &nAlt, altLen, altArcs));
               htkamd_trans *head = NULL;
               for (int i = 0; i < nAlt; i++) {
                  htkamd_trans *tr = NULL;
                  if (alMode) CHECK(htkamd_lattice_align_trans(&lat, &la, mmf, net, altArcs + (size_t)i * maxWords, altLen[i], &tr));      /* model / state labels from the arcs' records */
                  if (tr) { if (!head) head = tr; else CHECK(htkamd_trans_append_alternative(head, tr)); continue; }
                  CHECK(htkamd_trans_create(0, &tr));
                  for (int j = 0; j < altLen[i]; j++) {
                     const int arc = altArcs[(size_t)i * maxWords + j], pron = lat.nodePron[lat.arcEnd[arc]];
                     const char *sym = pron >= 0 ? htkamd_net_out_sym(net, pron) : NULL;
                     if (!sym || !sym[0]) continue;
                     CHECK(htkamd_trans_add(tr, (double)lat.nodeFrame[lat.arcStart[arc]] * ob.period, (double)lat.nodeFrame[lat.arcEnd[arc]] * ob.period, sym,
                                            htkamd_lattice_arc_score(&lat, arc), NULL, 0, NULL, 0));
                  }
                  if (!head) head = tr; else CHECK(htkamd_trans_append_alternative(head, tr));
               }
               if (head) {
                  CHECK(htkamd_trans_format(head, (double)ob.period, alMode ? states : 0, alMode ? models : 0, oflags));
                  make_fn(fn, outDir, outExt, out, sizeof(out));
                  if (mout) CHECK(htkamd_mlf_out_add(mout, out, head)); else CHECK(htkamd_trans_write(head, out));
                  htkamd_trans_free(head);
               }
               free(altLen); free(altArcs);
            }
            free(nN); free(nA); free(nodeFrame); free(nodePron); free(nodeLike); free(aS); free(aE); free(aAc); free(aLm); free(aPr); free(fo);
            free(ao.arcAlignOff); free(ao.alState); free(ao.alModel); free(ao.alDur); free(ao.alLike);
         }
         free_observations(&ob);
         continue;
      }
      int *nWords = (int *)malloc(sizeof(int) * (size_t)count), *wPron = (int *)malloc(sizeof(int) * (size_t)count * maxWords);
      int *wStart = (int *)malloc(sizeof(int) * (size_t)count * maxWords), *wEnd = (int *)malloc(sizeof(int) * (size_t)count * maxWords);
      float *wScore = (float *)malloc(sizeof(float) * (size_t)count * maxWords), *wLm = (float *)malloc(sizeof(float) * (size_t)count * maxWords);
      double *total = (double *)malloc(sizeof(double) * (size_t)count);
      CHECK(htkamd_decoder_run(udec, &dc, ob.dX, ob.frameOff, count, maxWords, nWords, wPron, wStart, wEnd, wScore, wLm, total, NULL));
      float usedBeam = dc.genBeam;
      if (align && genBeamInc > 0.0f) {                       /* DoAlignment's retries (HVite.c:900-913): wider beams while nothing reaches the final node */
         htkamd_decode_config dr = dc;
         float cur = dc.genBeam + genBeamInc;
         while (nWords[0] < 0 && cur <= genBeamLim - genBeamInc) {
            if (trace & 1) printf("No tokens survived to final node of network at beam %.1f\n", cur - genBeamInc);
            dr.genBeam = usedBeam = cur;
            CHECK(htkamd_decoder_run(udec, &dr, ob.dX, ob.frameOff, count, maxWords, nWords, wPron, wStart, wEnd, wScore, wLm, total, NULL));
            cur += genBeamInc;
         }
         if (nWords[0] < 0) { dr.genBeam = usedBeam = cur; CHECK(htkamd_decoder_run(udec, &dr, ob.dX, ob.frameOff, count, maxWords, nWords, wPron, wStart, wEnd, wScore, wLm, total, NULL)); }
      }
      for (int u = 0; u < count; u++) {
         const char *fn = files.v[first + u];
         const int T = ob.frameOff[u + 1] - ob.frameOff[u];
         if (nWords[u] < 0) { fprintf(stderr, "No tokens survived to final node of network: %s\n", fn); continue; }
         htkamd_trans *tr; CHECK(htkamd_trans_create((models ? 1 : 0) + (states ? 1 : 0), &tr));
         emit(tr, unet, mmf, vit, ob.dX, D, ob.frameOff[u], T, nWords[u], wPron + (size_t)u * maxWords, wStart + (size_t)u * maxWords, wEnd + (size_t)u * maxWords,
              wScore + (size_t)u * maxWords, wLm + (size_t)u * maxWords, (double)ob.period, models, states, lmScale, wordPen, usedBeam);
         CHECK(htkamd_trans_format(tr, (double)ob.period, states, models, oflags));
         char out[2048];
         make_fn(fn, outDir, outExt, out, sizeof(out));
         if (mout) CHECK(htkamd_mlf_out_add(mout, out, tr)); else CHECK(htkamd_trans_write(tr, out));
         htkamd_trans_free(tr);
         if (trace & 1) printf("File: %s\n==  [%d frames] %.4f [Ac=%.1f]\n", fn, T, total[u] / T, total[u]);
      }
      free(nWords); free(wPron); free(wStart); free(wEnd); free(wScore); free(wLm); free(total);
      if (align) { htkamd_decoder_destroy(udec); htkamd_net_destroy(unet); }
      free_observations(&ob);
   }
   if (mout) htkamd_mlf_out_close(mout);
   if (mlf) htkamd_mlf_free(mlf);
   if (dec) htkamd_decoder_destroy(dec);
   if (net) htkamd_net_destroy(net);
   htkamd_viterbi_destroy(vit); htkamd_model_destroy(model); htkamd_mmf_destroy(mmf);
   return 0;
}
