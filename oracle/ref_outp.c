/* ref_outp.c -- harness around the REFERENCE state output probability OutP (TEST INFRASTRUCTURE).
 *
 * Compiled by oracle/Makefile against the reference's headers and oracle/_ref/HTKLib.a.  Loads a model set and one parameter file
 * and writes OutP(&obs, hmm, j) (HModel.h:560 -> POutP -> SOutP -> MOutP: HModel.c:5503-5600) for every frame, every physical model
 * of the list and every emitting state, as raw floats -- with the set as it was loaded (DIAGC variances, linear weights: DOutP and
 * MixLogWeight on the fly, which is what HRest / HInit see), or after ConvDiagC + ConvLogWt (`-c`: IDOutP, what HERest / HVite see).
 * The vectors pin oracle/htk_oracle.c's orc_soutp_block / orc_doutp.
 *
 * usage: ref_outp [-c] -H mmf hmmlist datafile out.bin        out.bin: float[T][H][maxEmit] (native endian; 0 where a model is shorter)
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
#include "HModel.h"
#include "HUtil.h"

int main(int argc, char *argv[])
{
   HMMSet hset;
   MemHeap hmmStack, dataStack;
   char *s, *hmmList, *datafn, *outfn;
   Boolean conv = FALSE, eSep;
   ParmBuf pbuf;
   BufferInfo info;
   Observation obs;
   FILE *f;
   int T, t, h, j, maxEmit = 0;
   HMMScanState hss;

   if (InitShell(argc, argv, "ref_outp", "") < SUCCESS) HError(9999, "InitShell");
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
      else HError(9999, "unknown switch %s", s);
   }
   hmmList = GetStrArg(); datafn = GetStrArg(); outfn = GetStrArg();
   if (MakeHMMSet(&hset, hmmList) < SUCCESS || LoadHMMSet(&hset, NULL, NULL) < SUCCESS) HError(9999, "loading the model set failed");
   if (conv) { ConvDiagC(&hset, TRUE); ConvLogWt(&hset); }
   if ((pbuf = OpenBuffer(&dataStack, datafn, 0, UNDEFF, FALSE_dup, FALSE_dup)) == NULL) HError(9999, "OpenBuffer");
   GetBufferInfo(pbuf, &info);
   SetStreamWidths(info.tgtPK, info.tgtVecSize, hset.swidth, &eSep);
   obs = MakeObservation(&gstack, hset.swidth, info.tgtPK, FALSE, eSep);
   T = ObsInBuffer(pbuf);
   NewHMMScan(&hset, &hss);
   do { if (hss.hmm->numStates - 2 > maxEmit) maxEmit = hss.hmm->numStates - 2; } while (GoNextHMM(&hss));
   EndHMMScan(&hss);
   if ((f = fopen(outfn, "wb")) == NULL) HError(9999, "cannot create %s", outfn);
   for (t = 0; t < T; t++) {
      FILE *lf = fopen(hmmList, "r");
      char name[256];
      ReadAsTable(pbuf, t, &obs);
      for (h = 0; fscanf(lf, "%255s", name) == 1; h++) {
         MLink ml = FindMacroName(&hset, 'l', GetLabId(name, FALSE));
         HLink hmm;
         if (ml == NULL) HError(9999, "model %s not in the set", name);
         hmm = (HLink)ml->structure;
         for (j = 2; j < 2 + maxEmit; j++) {
            float v = (j < hmm->numStates) ? OutP(&obs, hmm, j) : 0.0f;
            fwrite(&v, sizeof(float), 1, f);
         }
      }
      fclose(lf);
   }
   fclose(f);
   return 0;
}
