/* ref_fbdump.c -- harness around the REFERENCE forward-backward (TEST INFRASTRUCTURE).
 *
 * Compiled by oracle/Makefile against the reference sources where they lie: it #includes the
 * reference's HFB.c (include path points at $(REF)/HTKLib) so that the file-static functions
 * StepBack / StepAlpha / SetOcct / UpMixParms / UpTranParms and the AlphaBeta internals are
 * reachable, links the rest from oracle/_ref/HTKLib.a, and dumps raw binary alpha/beta/
 * output-prob/occupation values per utterance plus the accumulators (DumpAccs) -- the vectors
 * the C restatement (htk_oracle.c) and the HIP path are checked against.  No reference source
 * text is copied into this file; the driver below mirrors what HERest.c:569-645,753-787 calls.
 *
 * usage: ref_fbdump [-C cfg] [-t f [i l]] [-c f] -H mmf -L labdir [-p N -M accdir] hmmlist dump.bin data...
 * dump.bin (native endian):
 *   int32 nUtt; then per utterance:
 *   int32 ok,T,Q,maxN; double pr; int32 qLo[T],qHi[T],aLo[T],aHi[T];
 *   double beta[T*Q*maxN] (NaN = NULL); double alpha[T*Q*maxN]; float outp[T*Q*maxN] (NaN = n/a);
 *   float occ[T*Q*maxN] (NaN = outside alpha beam)
 */
#include "HFB.c"          /* the reference file itself, via -I$(REF)/HTKLib */

static MemHeap hmmStack, uttStack, fbInfoStack, accStack;
static XFInfo xfInfo;

static void wr(FILE *f, const void *p, size_t n) { if (fwrite(p, 1, n, f) != n) HError(9999, "write"); }

int main(int argc, char *argv[])
{
   HMMSet hset;
   UttInfo *utt;
   FBInfo *fbInfo;
   char *s, *labDir = NULL, *accDir = NULL, *dumpfn, *hmmList;
   LogDouble pruneInit = NOPRUNE, pruneInc = 0.0, pruneLim = NOPRUNE;
   float minFrwdP = NOPRUNE;
   UPDSet uFlags = (UPDSet)(UPMEANS | UPVARS | UPTRANS | UPMIXES);
   int parMode = 1, nUtt = 0, firstTime = 1, totalT = 0;
   LogDouble totalPr = 0;
   FILE *df;
   long pos0;

   if (InitShell(argc, argv, "ref_fbdump", "") < SUCCESS) HError(9999, "InitShell");
   InitMem(); InitMath(); InitSigP(); InitAudio(); InitWave(); InitVQ(); InitLabel(); InitModel();
   if (InitParm() < SUCCESS) HError(9999, "InitParm");
   InitTrain(); InitUtil(); InitFB(); InitAdapt(&xfInfo); InitMap();
   CreateHeap(&hmmStack, "HmmStore", MSTAK, 1, 1.0, 50000, 500000);
   CreateHMMSet(&hset, &hmmStack, TRUE);
   CreateHeap(&uttStack, "uttStore", MSTAK, 1, 0.5, 100, 1000);
   utt = (UttInfo *)New(&uttStack, sizeof(UttInfo));
   CreateHeap(&fbInfoStack, "FBInfoStore", MSTAK, 1, 0.5, 100, 1000);
   fbInfo = (FBInfo *)New(&fbInfoStack, sizeof(FBInfo));
   CreateHeap(&accStack, "accStore", MSTAK, 1, 1.0, 50000, 500000);

   while (NextArg() == SWITCHARG) {
      s = GetSwtArg();
      switch (s[0]) {
      case 't':
         pruneInit = GetChkedFlt(0.0, 1.0E20, s);
         if (NextArg() == FLOATARG || NextArg() == INTARG) {
            pruneInc = GetChkedFlt(0.0, 1.0E20, s);
            pruneLim = GetChkedFlt(0.0, 1.0E20, s);
         } else { pruneInc = 0.0; pruneLim = pruneInit; }
         break;
      case 'c': minFrwdP = GetChkedFlt(0.0, 1000.0, s); break;
      case 'H': AddMMF(&hset, GetStrArg()); break;
      case 'L': labDir = GetStrArg(); break;
      case 'M': accDir = GetStrArg(); break;
      case 'p': parMode = GetChkedInt(0, 500, s); break;
      default: HError(9999, "ref_fbdump: unknown switch %s", s);
      }
   }
   hmmList = GetStrArg();
   dumpfn = GetStrArg();

   /* HERest.c:569-645 Initialise */
   if (MakeHMMSet(&hset, hmmList) < SUCCESS) HError(9999, "MakeHMMSet");
   if (LoadHMMSet(&hset, NULL, NULL) < SUCCESS) HError(9999, "LoadHMMSet");
   AttachAccs(&hset, &accStack, uFlags);
   ZeroAccs(&hset, uFlags);
   ConvDiagC(&hset, TRUE);
   InitialiseForBack(fbInfo, &fbInfoStack, &hset, uFlags, pruneInit, pruneInc, pruneLim, minFrwdP);
   ConvLogWt(&hset);
   InitUttInfo(utt, FALSE);
   fbInfo->inXForm = NULL; fbInfo->al_inXForm = NULL; fbInfo->paXForm = NULL;

   df = fopen(dumpfn, "wb");
   if (!df) HError(9999, "cannot open %s", dumpfn);
   pos0 = ftell(df);
   wr(df, &nUtt, 4);

   while (NumArgs() > 0) {
      char *datafn = GetStrArg();
      int ok, T, Q, maxN = 0, q, t, i, start, end;
      AlphaBeta *ab;
      double nan_d = NAN; float nan_f = NAN;
      double *beta, *alpha; float *outp, *occ; int *qLo, *qHi, *aLo, *aHi;
      size_t n;

      /* HERest.c:753-787 DoForwardBackward */
      utt->twoDataFiles = FALSE;
      utt->S = fbInfo->al_hset->swidth[0];
      LoadLabs(utt, UNDEFF, datafn, labDir, "lab");
      LoadData(fbInfo->al_hset, utt, UNDEFF, datafn, NULL);
      if (firstTime) { InitUttObservations(utt, fbInfo->al_hset, datafn, fbInfo->maxMixInS); firstTime = 0; }

      ok = StepBack(fbInfo, utt, datafn) ? 1 : 0;           /* HFB.c:1927 */
      T = utt->T; Q = utt->Q; ab = fbInfo->ab;
      wr(df, &ok, 4); wr(df, &T, 4); wr(df, &Q, 4);
      if (!ok) { maxN = 0; wr(df, &maxN, 4); nUtt++; ResetStacks(ab); continue; }
      for (q = 1; q <= Q; q++) if (ab->al_qList[q]->numStates > maxN) maxN = ab->al_qList[q]->numStates;
      wr(df, &maxN, 4); wr(df, &utt->pr, 8);
      n = (size_t)T * Q * maxN;
      beta = malloc(n * 8); alpha = malloc(n * 8); outp = malloc(n * 4); occ = malloc(n * 4);
      qLo = malloc(T * 4); qHi = malloc(T * 4); aLo = malloc(T * 4); aHi = malloc(T * 4);
      for (i = 0; i < (int)n; i++) { beta[i] = nan_d; alpha[i] = nan_d; outp[i] = nan_f; occ[i] = nan_f; }
      for (t = 1; t <= T; t++) {
         PruneInfo *p = ab->pInfo;
         int lo, hi;
         qLo[t - 1] = p->qLo[t]; qHi[t - 1] = p->qHi[t];
         /* CreateBetaQ (HFB.c:807) allocates [qLo-2 .. qHi+1] of the INITIAL beam; probe only what is
            certainly inside: the final beam plus one above */
         lo = p->qLo[t]; hi = p->qHi[t];
         for (q = lo; q <= hi; q++) {
            DVector b = ab->beta[t][q];
            if (b != NULL)
               for (i = 1; i <= ab->al_qList[q]->numStates; i++)
                  beta[((size_t)(t - 1) * Q + (q - 1)) * maxN + (i - 1)] = b[i];
            if (ab->otprob[t][q] != NULL)
               for (i = 2; i < ab->al_qList[q]->numStates; i++)
                  outp[((size_t)(t - 1) * Q + (q - 1)) * maxN + (i - 1)] = ab->otprob[t][q][i][0][0];
         }
      }
      /* HFB.c:1752-1810 StepForward, with dumps */
      CreateAlpha(ab, fbInfo->al_hset, utt->Q);
      InitAlpha(ab, &start, &end, utt->Q, fbInfo->skipstart, fbInfo->skipend);
      ab->occa = NULL;
      for (q = 1; q <= utt->Q; q++) {
         HLink up_hmm = ab->up_qList[q];
         long negs = (long)up_hmm->hook + 1;
         up_hmm->hook = (void *)negs;
      }
      ResetObsCache();
      for (t = 1; t <= utt->T; t++) {
         GetInputObs(utt, t, fbInfo->hsKind);
         if (t > 1)
            StepAlpha(ab, t, &start, &end, utt->Q, utt->T, utt->pr, fbInfo->skipstart, fbInfo->skipend);
         aLo[t - 1] = start; aHi[t - 1] = end;
         for (q = 1; q <= Q; q++)
            for (i = 1; i <= ab->al_qList[q]->numStates; i++)
               alpha[((size_t)(t - 1) * Q + (q - 1)) * maxN + (i - 1)] = ab->alphat[q][i];
         for (q = start; q <= end; q++) {
            HLink al_hmm = ab->al_qList[q], up_hmm = ab->up_qList[q];
            DVector aqt = ab->alphat[q];
            DVector bqt = ab->beta[t][q];
            DVector bqt1 = (t == utt->T) ? NULL : ab->beta[t + 1][q];
            DVector aqt1 = (t == 1) ? NULL : ab->alphat1[q];
            DVector bq1t = (q == utt->Q) ? NULL : ab->beta[t][q + 1];
            SetOcct(al_hmm, q, ab->occt, ab->occa, aqt, bqt, bq1t, utt->pr);
            for (i = 1; i <= al_hmm->numStates; i++)
               occ[((size_t)(t - 1) * Q + (q - 1)) * maxN + (i - 1)] = ab->occt[i];
            if (fbInfo->uFlags & (UPMEANS | UPVARS | UPMIXES | UPXFORM))
               UpMixParms(fbInfo, q, up_hmm, al_hmm, utt->ot, utt->ot2, t, aqt, aqt1, bqt,
                          utt->S, utt->twoDataFiles, utt->pr);
            if (fbInfo->uFlags & UPTRANS)
               UpTranParms(fbInfo, up_hmm, t, q, aqt, bqt, bqt1, bq1t, utt->pr);
         }
      }
      wr(df, qLo, T * 4); wr(df, qHi, T * 4); wr(df, aLo, T * 4); wr(df, aHi, T * 4);
      wr(df, beta, n * 8); wr(df, alpha, n * 8); wr(df, outp, n * 4); wr(df, occ, n * 4);
      free(beta); free(alpha); free(outp); free(occ); free(qLo); free(qHi); free(aLo); free(aHi);
      totalT += utt->T; totalPr += utt->pr;
      ResetStacks(ab);
      nUtt++;
   }
   fseek(df, pos0, SEEK_SET); wr(df, &nUtt, 4); fclose(df);

   if (accDir != NULL) {     /* HERest.c:543-550 */
      char newFn[MAXSTRLEN]; FILE *f; float tmpFlt;
      MakeFN("HER$.acc", accDir, NULL, newFn);
      f = DumpAccs(&hset, newFn, uFlags, parMode);
      tmpFlt = (float)totalPr;
      WriteFloat(f, &tmpFlt, 1, TRUE);
      WriteInt(f, (int *)&totalT, 1, TRUE);
      fclose(f);
   }
   return 0;
}
