/* ref_outpblock.c -- harness for shim/hlvmodel_outp_shim.c (TEST INFRASTRUCTURE): HDecode's block scorer interface against the
 * reference's own OutP.
 *
 * Compiled by oracle/Makefile against the reference's headers (HTKLib + HTKLVRec/HLVModel.h), linked with oracle/_ref/HTKLib.a, the
 * shim and libhtk_amd.so.  Loads a model set and a parameter file, makes the StateInfo_lv HDecode would hand to its scorer (USEHMODEL
 * = T form: the HModel StateInfo pointers, numbered in scan order), then walks the file the way cOutP does -- a block of `n` frames
 * ahead per state (HLVRec-outP.c:196-262) -- through OutPBlock_HMod, and compares every score with acScale * OutP(&obs, hmm, j)
 * (HModel.h:560), the value OutPBlock_HMod computes in the reference (HLVRec-outP.c:329).
 *   usage: ref_outpblock [-c] [-H mmf] [-d dir] hmmlist datafile blockSize acScale       -c: ConvDiagC + ConvLogWt first, as HDecode's set is
 * prints "scores N mismatches M maxdiff X" and exits 0 when M == 0.
 */
#include "HShell.h"
#include "HMem.h"
#include "HMath.h"
#include "HSigP.h"
#include "HAudio.h"
#include "HWave.h"
#include "HVQ.h"
#include "HParm.h"
#include "HLabel.h"
#include "HDict.h"
#include "HModel.h"
#include "HUtil.h"
#include "HLVModel.h"

void OutPBlock_HMod(StateInfo_lv *si, Observation **obsBlock, int n, int sIdx, float acScale, LogFloat *outP, int id);

int main(int argc, char *argv[])
{
   HMMSet hset;
   MemHeap hmmStack, dataStack;
   char *s, *hmmList, *datafn, *hmmDir = NULL;
   Boolean conv = FALSE, eSep;
   ParmBuf pbuf;
   BufferInfo info;
   Observation *obs, **blk;
   StateInfo_lv si;
   HMMScanState hss;
   int T, t, i, nS = 0, block, N = 0, bad = 0;
   float acScale, out[64];
   double maxd = 0.0;

   if (InitShell(argc, argv, "ref_outpblock", "") < SUCCESS) HError(9999, "InitShell");
   InitMem(); InitMath(); InitSigP(); InitAudio(); InitWave(); InitVQ(); InitLabel(); InitModel();
   if (InitParm() < SUCCESS) HError(9999, "InitParm");
   InitUtil();
   CreateHeap(&hmmStack, "HmmStore", MSTAK, 1, 1.0, 50000, 500000);
   CreateHeap(&dataStack, "dataStore", MSTAK, 1, 0.5, 1000, 10000);
   CreateHMMSet(&hset, &hmmStack, TRUE);
   while (NextArg() == SWITCHARG) {
      s = GetSwtArg();
      if (s[0] == 'c') conv = TRUE;
      else if (s[0] == 'H') AddMMF(&hset, GetStrArg());
      else if (s[0] == 'd') hmmDir = GetStrArg();
      else HError(9999, "unknown switch %s", s);
   }
   hmmList = GetStrArg(); datafn = GetStrArg(); block = GetIntArg(); acScale = GetFltArg();
   if (block < 1 || block > 64) HError(9999, "block size 1..64");
   if (MakeHMMSet(&hset, hmmList) < SUCCESS || LoadHMMSet(&hset, hmmDir, NULL) < SUCCESS) HError(9999, "loading the model set failed");
   if (conv) { ConvDiagC(&hset, TRUE); ConvLogWt(&hset); }
   if ((pbuf = OpenBuffer(&dataStack, datafn, 0, UNDEFF, FALSE_dup, FALSE_dup)) == NULL) HError(9999, "OpenBuffer");
   GetBufferInfo(pbuf, &info);
   SetStreamWidths(info.tgtPK, info.tgtVecSize, hset.swidth, &eSep);
   T = ObsInBuffer(pbuf);
   obs = (Observation *)New(&gstack, sizeof(Observation) * (T + block));
   for (t = 0; t < T + block; t++) {
      obs[t] = MakeObservation(&gstack, hset.swidth, info.tgtPK, FALSE, eSep);
      ReadAsTable(pbuf, t < T ? t : T - 1, &obs[t]);                /* the last frame repeated past the end, as a decoder pads its block */
   }
   blk = (Observation **)New(&gstack, sizeof(Observation *) * block);
   /* the scorer's view of the set */
   memset(&si, 0, sizeof(si));
   si.hset = &hset; si.useHModel = TRUE; si.nDim = hset.vecSize;
   NewHMMScan(&hset, &hss);
   while (GoNextState(&hss, FALSE)) hss.si->sIdx = nS++;
   EndHMMScan(&hss);
   si.si = (StateInfo **)New(&gstack, sizeof(StateInfo *) * nS);
   NewHMMScan(&hset, &hss);
   while (GoNextState(&hss, FALSE)) si.si[hss.si->sIdx] = hss.si;
   EndHMMScan(&hss);
   for (t = 0; t < T; t += block) {
      int sIdx;
      for (i = 0; i < block; i++) blk[i] = &obs[t + i];
      for (sIdx = 0; sIdx < nS; sIdx++) {
         OutPBlock_HMod(&si, blk, block, sIdx, acScale, out, t);
         for (i = 0; i < block; i++) {
            const float want = POutP(&hset, blk[i], si.si[sIdx]) * acScale;
            const double dd = fabs((double)want - (double)out[i]);
            N++;
            if (want != out[i]) bad++;
            if (dd > maxd) maxd = dd;
         }
      }
   }
   printf("states %d frames %d scores %d mismatches %d maxdiff %g\n", nS, T, N, bad, maxd);
   return bad ? 1 : 0;
}
