/* htk_oracle.h -- CPU restatement of the reference HTK hot path (TEST INFRASTRUCTURE).
 *
 * This is the parity ORACLE: plain C that follows the reference's arithmetic (types, operation
 * order, thresholds) for the HERest / HVite hot path, each function citing the reference
 * file:line it restates.  It is pinned against the reference itself (oracle/_ref built from
 * /root/reference by oracle/Makefile; see tests/test_oracle_vs_ref.py and tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (htk_amd/, include/) never links, imports or calls it.
 *
 * Indexing: model arrays are 0-based; HMM states keep the reference's numbering 1..N
 * (1 = entry, N = exit) so that formulas read like the reference.
 */
#ifndef HTK_ORACLE_H
#define HTK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LZERO   (-1.0E10)    /* HMath.h:42 */
#define ORC_LSMALL  (-0.5E10)    /* HMath.h:43 */
#define ORC_MINEARG (-708.3)     /* HMath.h:44 */
#define ORC_MINLARG 2.45E-308    /* HMath.h:45 */
#define ORC_TPI     6.28318530717959   /* HMath.h:41 (truncated literal, as in the reference) */
#define ORC_PI      3.14159265358979   /* HMath.h:40 */
#define ORC_MINMIX  1.0E-5       /* HModel.h:52 */
#define ORC_LMINMIX (-11.5129254649702) /* HModel.h:53 */
#define ORC_NOPRUNE 1.0E20       /* HFB.h:31 */

/* update flags, HTrain.h / HModel.h UPDSet bits actually used on the path */
#define ORC_UPMEANS 1
#define ORC_UPVARS  2
#define ORC_UPTRANS 4
#define ORC_UPMIXES 8

typedef struct {
   int D;                     /* vecSize */
   int S;                     /* tied states (StateInfo with sIdx)        */
   int C;                     /* mixture components = sum_s nMix(s)       */
   int G;                     /* distinct Gaussians (MixPDF with mIdx)    */
   int nT;                    /* transition matrices                       */
   int H;                     /* physical HMMs                             */
   const int   *stateCompOff; /* [S+1] components of state s               */
   const float *compLogWt;    /* [C] MixLogWeight() value (HModel.c:5288)  */
   const int   *compGauss;    /* [C] Gaussian index of component           */
   const float *mean;         /* [G*D]                                     */
   const float *ivar;         /* [G*D] after ConvDiagC (HUtil.c:413)       */
   const float *gconst;       /* [G]   DIAGC gConst (HModel.c:5641)        */
   const int   *transN;       /* [nT] states per matrix                    */
   const int   *transOff;     /* [nT+1] offset of matrix t in transP       */
   const float *transP;       /* log probs row-major N*N, LZERO = none     */
   const int   *hmmTrans;     /* [H]                                       */
   const int   *hmmStateOff;  /* [H+1]                                     */
   const int   *hmmState;     /* tied-state index of emitting states 2..N-1*/
   /* several streams (hset->swidth[0] = NSt > 1; 0 or 1: one stream).  stateCompOff then has S*NSt + 1 entries -- the components of
      stream k of state s are [s*NSt + k] .. [s*NSt + k + 1) -- and a Gaussian is held in an undivided row of D elements of which
      only the dimensions d with dimStream[d] == its stream are read: a stream vector is those elements in ascending order
      (ExtractObservation HParm.c:2843).  wtOcc accumulators: [S*NSt]. */
   int NSt;
   const int   *dimStream;    /* [D] */
   /* tied-mixture sets (hsKind TIEDHS, <TMIX>): every (state, stream) lists the pool of its stream (compGauss equal across states);
      the arithmetic is PrecomputeTMix / SOutP's (HModel.c:5308,5555) and UpMixParms' TIEDHS branches (HFB.c:1524-1600): it needs the
      LINEAR weights (tpdf) and the VARIANCES (the set stays DIAGC: ConvDiagC / ConvLogWt skip it, HUtil.c:419,478) */
   int tiedMix;
   const float *compWeight;   /* [C] linear weights */
   const float *var;          /* [G*D] variances */
   int msIntended;            /* 0: Setotprob as the reference has it, with its second-visit branch (HFB.c:1044,1059); 1: every visit
                                 computes the first visit's values (what that branch equals for S = 3 only) */
} orc_model;

typedef struct {              /* float accumulators exactly as HTrain.h:211-232 */
   float *mu;                 /* [G*D] MuAcc.mu   */
   float *muOcc;              /* [G]   MuAcc.occ  */
   float *va;                 /* [G*D] VaAcc.cov.var */
   float *vaOcc;              /* [G]   VaAcc.occ  */
   float *wt;                 /* [C]   WtAcc.c    */
   float *wtOcc;              /* [S]   WtAcc.occ  */
   float *tr;                 /* [transOff[nT]] TrAcc.tran */
   float *trOcc;              /* [sum_t N_t] TrAcc.occ, matrix t at sum_{k<t} N_k */
   int   *nEgs;               /* [H] hmm->hook example counter (HFB.c:1768-1772) */
} orc_accs;

typedef struct {
   double pruneInit, pruneInc, pruneLim;   /* HFB.c:76-83 pruneSetting */
   float  minFrwdP;
   int    uFlags;
} orc_fbcfg;

typedef struct {              /* optional dumps; any pointer may be NULL */
   double *beta;              /* [T*Q*maxN] NaN where the reference holds NULL */
   double *alpha;             /* [T*Q*maxN] alphat column after step t          */
   float  *outp;              /* [T*Q*maxN] state output prob, NaN if not evaluated */
   int    *qLo, *qHi;         /* [T] final beta beam                            */
   int    *aLo, *aHi;         /* [T] alpha beam                                  */
   float  *occ;               /* [T*Q*maxN] occt                                 */
   long long nEval;           /* number of (t, chain state) output-prob evaluations (Setotprob) */
} orc_fbdump;

/* ---- log arithmetic (HMath.c:1576) ---- */
double orc_ladd(double x, double y);

/* ---- model preparation ---- */
void  orc_fix_diag_gconst(int D, const float *var, float *gconst_out);      /* HModel.c:5641 */
void  orc_fix_diag_gconst_ms(int D, const float *var, const int *dimStream, int stream, float *gconst_out);   /* the same over one stream's dimensions */
void  orc_conv_diagc(int n, const float *var, float *ivar_out);              /* HUtil.c:413   */
float orc_mix_log_weight(float w);                                           /* HModel.c:5288 */
int   orc_min_dur(int N, const float *transP);                               /* HFB.c:106     */

/* ---- GMM scoring ---- */
float orc_idoutp(const float *x, int D, const float *mean, const float *ivar, float gconst); /* HModel.c:5420 */
/* ShStrP (HFB.c:898) == cSOutP (HRec.c:438) arithmetic: returns state log-lik, fills mixp[0..M-1]
   (LZERO for skipped components) when mixp != NULL */
float orc_state_outp(const orc_model *m, int s, const float *x, float *mixp);
/* the same for stream element e = state*NSt + stream of a multi-stream set (x: the undivided row) */
float orc_elem_outp(const orc_model *m, int e, const float *x, float *mixp);
float orc_doutp(const float *x, int D, const float *mean, const float *var, float gconst);    /* HModel.c:5347 */
void  orc_soutp_block(const orc_model *m, const float *var, const float *X, int T, const int *states, int ns, float *out);
void  orc_score_block_diagc(const orc_model *m, const float *var, const float *X, int T, const int *states, int ns, float *out);
/* SOutP arithmetic (HModel.c:5503): double accumulation, one float rounding */
float orc_soutp(const orc_model *m, int s, const float *x);
/* dense block: out[t*ns + k] = orc_state_outp(states[k], X[t]) */
void  orc_score_block(const orc_model *m, const float *X, int T, const int *states, int ns, float *out);

/* ---- forward-backward for one utterance (FBFile, HFB.c:1923) ---- */
/* returns 1 on success (stats accumulated, *pr set), 0 if the utterance is skipped (-7324),
   <0 on the reference's fatal errors (-7332 tee-model placement, -7390 alpha prune failure) */
int orc_fb_utt(const orc_model *m, const orc_fbcfg *cfg, const float *X, int T,
               const int *labs, int Q, orc_accs *acc, double *pr, orc_fbdump *dump);

/* ---- model update (HERest.c:1262 MLUpdateModels) on packed arrays, in place ---- */
typedef struct {
   int   minEgs;              /* HERest.c:96 default 3 */
   float minVar;              /* -v, HERest.c:95 default 0.0 */
   float mixWeightFloor;      /* -w f  => f*MINMIX, HERest.c:425 */
   int   uFlags;
   int   singleProcess;       /* parMode == -1: ForceDiagC + ConvExpWt round trips first (HERest.c:1336-1339) */
} orc_updcfg;
typedef struct {
   int nFloorVar, nFloorVarMix;  /* HERest.c:791-792 */
   int nSkippedHmm;              /* models with < minEgs examples */
} orc_updstats;
void orc_update(const orc_model *m, const orc_accs *acc, const orc_updcfg *cfg,
                float *mean /*[G*D] in/out*/, float *var /*[G*D] DIAGC in/out*/,
                float *gconst /*[G] in/out*/, float *compWeight /*[C] linear in/out*/,
                float *transP /*in/out (log)*/, orc_updstats *st);

/* ---- waveform -> MFCC front end (orc_mfcc.c): HWave.c:1575,1663,1683; HParm.c:2214 ConvertFrame, :1618 AddQualifiers;
        HSigP.c PreEmphasise :134, Ham :122, FFT :311, Realft :362, InitFBank :471, Wave2FBank :558, FBank2MFCC :607,
        FBank2C0 :647, WeightCepstrum :773, Regress :827, FZeroMean :803, NormaliseLogEnergy :911 ---- */
typedef struct {
   double sampPeriod;          /* SOURCERATE, 100 ns units (625 = 16 kHz) */
   double winDur, frPeriod;    /* WINDOWSIZE, TARGETRATE, 100 ns units     */
   int    numChans, numCeps, cepLifter;            /* NUMCHANS NUMCEPS CEPLIFTER */
   float  preEmph;                                 /* PREEMCOEF */
   int    useHam, usePower, zMeanSource, rawEnergy, eNormalise;   /* USEHAMMING USEPOWER ZMEANSOURCE RAWENERGY ENORMALISE */
   float  loFreq, hiFreq, cepScale, silFloor, eScale;             /* LOFREQ HIFREQ (<0 = off) CEPSCALE SILFLOOR ESCALE */
   int    hasC0, hasE, hasD, hasA, hasZ;           /* _0 _E _D _A _Z of TARGETKIND = MFCC... */
   int    delWin, accWin;                          /* DELTAWINDOW ACCWINDOW */
} orc_mfcc_cfg;
int orc_mfcc_frames(int nSamples, const orc_mfcc_cfg *c, int *frSize, int *frRate);
int orc_mfcc_cols(const orc_mfcc_cfg *c);
int orc_mfcc(const short *wav, int nSamples, const orc_mfcc_cfg *c, float *out);
int orc_add_qualifiers(const float *stat, int T, int nStat, int hasD, int hasA, int delWin, int accWin, float *out);
int orc_parm_qualify2(const float *stat, int T, int nStat, int nZeroMean, int hasD, int hasA, int hasT,
                      int delWin, int accWin, int thirdWin, int nullECol, int v1Compat, int simpleDiffs, float *out);
void orc_compv(const float *X, long T, int D, float minVar, float *mean, float *var);
int orc_parm_qualify(const float *stat, int T, int nStat, int nZeroMean, int hasD, int hasA, int hasT,
                     int delWin, int accWin, int thirdWin, int nullECol, float *out);

/* ---- Viterbi forced alignment of a chain of physical models (HRec token passing, 1-best; orc_viterbi.c) ----
   Returns the number of state segments (time order) or -1 when no token survives.  Frames are 0-based,
   [segStart, segEnd).  segScore = like(next Align record) - like(this one) (LatFromPaths HRec.c:1512). */
/* N-best token passing + lattice (orc_decode_n.c; HVite -n nToks): the lattice CreateLattice builds from the final token set.
   Node 0 = start, node 1 = end (no word); latNodeNet = network node of the word end (-1 start, -2 end).  Returns 0, -1 if no token
   reached the final node, -3 if the lattice does not fit. */
int orc_decode_nbest(const orc_model *m, const float *X, int T,
                     int nNodes, const int *kind, const int *model, const float *pronProb,
                     const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
                     float genBeam, float wordBeam, float nBeam, float lmScale, float wordPen, float prScale, int nToks, int maxActive,
                     int maxLatNodes, int maxLatArcs, int *latNodeNet, int *latNodeFrame, double *latNodeLike,
                     int *latArcStart, int *latArcEnd, float *latArcAc, float *latArcLm, float *latArcPr, double *latArcScore,
                     int *nLatNodes, int *nLatArcs, double *totalLike);
/* ... with alignment records (HVite -n with -m: alignMode & 1 / -f: & 2): arc j's lAlign = [arcAlignOff[j], arcAlignOff[j+1]) of alState /
   alNode (network node of the model) / alDur (frames) / alLike (LatFromPaths HRec.c:1582-1656, -DPHNALG) */
int orc_decode_nbest_align(const orc_model *m, const float *X, int T,
                     int nNodes, const int *kind, const int *model, const float *pronProb,
                     const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
                     float genBeam, float wordBeam, float nBeam, float lmScale, float wordPen, float prScale, int nToks, int maxActive, int alignMode,
                     int maxLatNodes, int maxLatArcs, int *latNodeNet, int *latNodeFrame, double *latNodeLike,
                     int *latArcStart, int *latArcEnd, float *latArcAc, float *latArcLm, float *latArcPr, double *latArcScore,
                     int *nLatNodes, int *nLatArcs, double *totalLike,
                     int maxAlign, int *arcAlignOff, int *alState, int *alNode, int *alDur, float *alLike);
/* 1-best decoding over a flat recognition network (orc_decode.c): returns the number of words, -1 if no token reached
   the final node, -3 if maxWords is too small, -4 if the zero-time nodes form a loop.  Frames are 0-based boundaries. */
int orc_decode(const orc_model *m, const float *X, int T,
               int nNodes, const int *kind, const int *model, const float *pronProb,
               const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
               float genBeam, float wordBeam, float lmScale, float wordPen, float prScale,
               int maxWords, int *wordPron, int *wordStart, int *wordEnd, float *wordScore, double *totalLike);
int orc_decode_u(const orc_model *m, const float *X, int T,
               int nNodes, const int *kind, const int *model, const float *pronProb,
               const int *linkOff, const int *linkDest, const float *linkLike, int initial, int final,
               float genBeam, float wordBeam, float lmScale, float wordPen, float prScale, int maxActive,
               int maxWords, int *wordPron, int *wordStart, int *wordEnd, float *wordScore, double *totalLike);
int orc_viterbi_align(const orc_model *m, const float *X, int T, const int *labs, int Q, float genBeam,
                      int maxSeg, int *segQ, int *segState, int *segStart, int *segEnd, double *segScore,
                      int *modStart, int *modEnd, double *modScore, double *totalLike);

#ifdef __cplusplus
}
#endif
#endif
