/* htk_amd.h -- C ABI of the MI355X-native HTK acoustic-scoring / Baum-Welch / Viterbi core.
 *
 * This is the drop-in boundary for the HERest / HVite hot path of HTK 3.4.1:
 * plain C, plain pointers and sizes, no C++ or torch types.  An HTKLib build binds these entry
 * points from the places cited on each declaration (INTEGRATION.md shows the shim a maintainer
 * adds to HModel.c / HFB.c / HRec.c / HParm.c).  Everything numeric runs in hand-written HIP
 * kernels for gfx950; there is no CPU fallback -- every call fails with HTKAMD_ENODEV when no
 * device is present.
 *
 * Conventions
 *  - "host" pointers are ordinary memory, "device" pointers are HBM addresses (hipMalloc or any
 *    allocator that yields device memory, e.g. a torch CUDA tensor's data_ptr()).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - All functions return 0 on success or a negative HTKAMD_E* code; htkamd_last_error() gives
 *    the message of the most recent failure on the calling thread.
 *  - Indices are 0-based; HMM state numbers inside a model keep HTK's 1..N (1 entry, N exit).
 */
#ifndef HTK_AMD_H
#define HTK_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HTKAMD_OK        0
#define HTKAMD_EINVAL   (-1)   /* bad argument */
#define HTKAMD_ENODEV   (-2)   /* no HIP device / kernel image not loadable */
#define HTKAMD_ENOMEM   (-3)
#define HTKAMD_EHIP     (-4)   /* HIP runtime error, see htkamd_last_error() */
#define HTKAMD_EMODEL   (-5)   /* model violates a restriction of this path (streams>1, non-diagonal cov...) */
#define HTKAMD_EIO      (-6)   /* file cannot be opened / read / written */
#define HTKAMD_ERANGE   (-7)   /* HTKAMD_SCORE_F16 only: a feature value or a model coefficient does not fit the fp16 scoring path's range;
                                  nothing of the pass can be used -- repeat it with HTKAMD_SCORE_BF16 (same tolerance class, fp32's exponent) */

/* HTK's log-arithmetic constants (HMath.h:42-45, HModel.h:52-53, HFB.h:31) */
#define HTKAMD_LZERO    (-1.0E10)
#define HTKAMD_LSMALL   (-0.5E10)
#define HTKAMD_NOPRUNE  1.0E20

/* UPDSet bits used by HERest (HModel.h UPDSet; HERest.c:97 default "tmvw") */
#define HTKAMD_UPMEANS  1
#define HTKAMD_UPVARS   2
#define HTKAMD_UPTRANS  4
#define HTKAMD_UPMIXES  8
#define HTKAMD_UPMAP    32   /* HERest -u p: MAP instead of ML re-estimation of means / variances / weights (HMap.c:413 MAPUpdateModels) */

int         htkamd_version(void);
const char *htkamd_last_error(void);
int         htkamd_device_count(void);
int         htkamd_set_device(int ordinal);

/* Device-memory plumbing for hosts that do not bring their own allocator (the C drivers, the tests). */
int htkamd_dev_malloc(void **dptr, size_t bytes);
int htkamd_dev_free(void *dptr);
int htkamd_memcpy_h2d(void *dDst, const void *hSrc, size_t bytes, void *stream);   /* synchronous on return */
/* page-locked host memory and a copy from it that does not wait: ordered before the work queued behind it on `stream`; refill the buffer only
   after that work has been waited for (the HFB shim stages an utterance's observations this way: shim/htklib_hfb_shim.c, FBFile HFB.c:1923) */
int htkamd_host_malloc(void **hptr, size_t bytes);
int htkamd_host_free(void *hptr);
int htkamd_memcpy_h2d_async(void *dDst, const void *hSrc, size_t bytes, void *stream);
int htkamd_memcpy_d2h(void *hDst, const void *dSrc, size_t bytes, void *stream);   /* synchronous on return */
int htkamd_stream_sync(void *stream);
/* streams for hosts without HIP headers (C drivers of the exchange in parts): a stream of one's own, and "the work queued on `waiter` from
   here on starts behind what has been queued on `signaller` so far" (an event recorded on the one, waited for by the other; NULL = the default stream) */
int htkamd_stream_create(void **stream);
int htkamd_stream_destroy(void *stream);
int htkamd_stream_wait(void *waiter, void *signaller);

/* ------------------------------------------------------------------------------------------
 * Packed HMM set.  Flat restatement of HMMSet / HMMDef / StateInfo / StreamElem / MixPDF
 * (HModel.h:85-152,297-340) for one stream, diagonal covariances, PLAINHS/SHAREDHS:
 *   tied states   <-> StateInfo with sIdx 1..S   (SetIndexes, HModel.c:3942)
 *   components    <-> MixtureElem {weight, mpdf}
 *   Gaussians     <-> MixPDF with mIdx 1..G      (several components may share one: ~m macros)
 *   transP        <-> SMatrix with tIdx, LOG probabilities, LZERO for "no transition"
 * Parameters are given as they stand after LoadHMMSet (DIAGC variances, linear weights);
 * the library applies FixDiagGConst (HModel.c:5641) when gconst==NULL, ConvDiagC (HUtil.c:413)
 * and ConvLogWt (HUtil.c:474) itself, exactly once, as HERest.c:592-645 / HVite.c:503 do.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
   int vecSize;               /* D  (hset->vecSize)                                  */
   int numStates;             /* S  tied states                                       */
   int numComp;               /* C  = sum over states of nMix                         */
   int numGauss;              /* G  distinct MixPDFs                                  */
   int numTrans;              /* nT distinct transition matrices                      */
   int numPhys;               /* H  physical HMMs                                     */
   const int   *stateCompOff; /* [S+1] components of state s: stateCompOff[s]..[s+1)  */
   const float *compWeight;   /* [C] linear mixture weights (MixtureElem.weight)      */
   const int   *compGauss;    /* [C] Gaussian index of each component                 */
   const float *mean;         /* [G*D] MixPDF.mean                                    */
   const float *var;          /* [G*D] MixPDF.cov.var, DIAGC (variances)              */
   const float *gconst;       /* [G] MixPDF.gConst or NULL to have it computed        */
   const int   *transN;       /* [nT] numStates N of each matrix (entry+exit included)*/
   const int   *transOff;     /* [nT+1] offset of matrix t inside transP              */
   const float *transP;       /* row-major N*N log probs per matrix                   */
   const int   *hmmTrans;     /* [H] tIdx of each physical HMM                        */
   const int   *hmmStateOff;  /* [H+1]                                                */
   const int   *hmmState;     /* tied-state index of the emitting states 2..N-1       */
   /* Several streams (hset->swidth[0] > 1; HModel.h StreamElem, HParm.c:3094 SetStreamWidths).  numStreams 0 or 1: one stream, the
      fields above as described.  numStreams = NS > 1: stateCompOff has numStates*NS + 1 entries, the components of stream k of state
      s are stateCompOff[s*NS + k] .. [s*NS + k + 1); a Gaussian belongs to one stream and is held in an UNDIVIDED row of vecSize
      elements: its values at the dimensions d with dimStream[d] == its stream, mean 0 and variance +infinity elsewhere (such a
      dimension contributes nothing to a score and is never re-estimated); the feature rows stay undivided as well.
      WHERE THIS LIBRARY DELIBERATELY DIFFERS FROM THE REFERENCE'S HERest (both differences are defects of the reference; the oracle
      restates them bit for bit, oracle/htk_oracle.c, and DESIGN.md §4b / §4c measures them):
        * NS != 3: when a tied state is met a second time in one frame (by another chain state), Setotprob sums the cached stream
          vectors it has already overwritten with "the sum of the others" and halves the result (HFB.c:1044-1064) -- the first visit's
          value for NS = 3 only; for NS = 2 every repeated state gets HALF its log probability (the reference reports -33.6 per frame
          on the demo set split 13 | 13 where the sum of the streams' log probabilities gives -59.1).  This library computes the
          first visit's value at every visit: equal to HERest for NS = 1 and NS = 3, NOT for NS = 2 or NS >= 4 on sets with repeated
          tied states -- unless htkamd_model_set_compat(HTKAMD_COMPAT_STREAM_REVISIT) asks for the reference's value.
        * SHAREDHS sets whose mixtures share pdfs (~m macros without HHEd's TIEDHS conversion): ConvLogWt converts the weights of the
          FIRST state that uses a shared pdf only (HUtil.c:474-485), the others are read as log weights; this library converts
          every weight once -- unless htkamd_model_set_compat(HTKAMD_COMPAT_SHARED_LOGWT) asks for the reference's reading. */
   int numStreams;
   const int   *dimStream;    /* [D] stream (0-based) of each dimension, NULL for one stream (htkamd_mmf computes it from the kind) */
   /* HTKAMD_HS_TIED: a tied-mixture set (hsKind TIEDHS, <TMIX>): every (state, stream) lists the SAME pool of Gaussians of its stream
      (compGauss equal across states) with its own weights; output probabilities and statistics go through the pool's top-M arithmetic
      (PrecomputeTMix HModel.c:5308, SOutP :5555, UpMixParms HFB.c:1524-1600), the pool is re-estimated once per set (HERest.c:1272). */
   int hsKind;
   const float *streamWeight; /* [numStates*NS] <SWEIGHTS> of every state, NULL = all 1 (GetStateInfo HModel.c:1994).  Used by the aligner and
                                 the decoder (OutP = sum_s w_s SOutP_s, HModel.c:5570-5583 / HRec.c:510-548); HFB ignores them (HFB.c:1057) */
} htkamd_model_desc;
#define HTKAMD_HS_PLAIN 0
#define HTKAMD_HS_TIED  1
/* dimStream for a parameter kind given as text ("MFCC_E_D") and the stream widths of <STREAMINFO>: the split SetStreamWidths /
   ExtractObservation make (HParm.c:3094,2843 -- the standard splits take the energy terms out into the last stream, any other split is
   consecutive pieces).  Returns 0, or -1 with the reason in `why`. */
int htkamd_host_stream_dims(const char *kind, int vecSize, int S, const int *width, int *dimStream /*[vecSize]*/, char *why, size_t whyLen);

typedef struct htkamd_model htkamd_model;

int  htkamd_model_create(const htkamd_model_desc *desc, htkamd_model **out);
/* Tied mean / variance vectors (~u / ~v macros: SVector with nUse > 1, HModel.c:1737-1790).  meanShare[g] / varShare[g] >= 0 name the
 * vector Gaussian g's mean / variance is a copy of (-1 = private; either array may be NULL); htkamd_mmf_sharing gives them for a set
 * read from files.  Scoring and the statistics are per Gaussian as always; htkamd_model_update pools the statistics of the sharers
 * as the reference's hooks on the shared vector do (one MuAcc / VaAcc per vector; a tied variance gets no mean-shift correction,
 * HERest.c:1080) and keeps the copies equal; htkamd_model_update_device does the same where the accumulators lie (the "first mixture
 * to reach the vector" of the reference's scan as a minimum over scan positions). */
int  htkamd_model_set_sharing(htkamd_model *m, const int *meanShare /*[G]*/, const int *varShare /*[G]*/);
/* The order in which UpdateModels (HERest.c:1262-1321) visits the physical models: the reference's HMM scan, i.e. htkamd_hmm_scan_order of
   their names.  It matters only for sets with mean vectors shared across models (which Gaussian's variance takes the mean-shift term);
   NULL (the default) = definition order. */
int  htkamd_model_set_scan_order(htkamd_model *m, const int *order /*[numPhys]*/);
int  htkamd_model_has_sharing(const htkamd_model *m);
/* Tied-mixture sets in the aligner and the decoders: the pruning threshold of the pool (HVite -c f, tmBeam HVite.c:115: pool entries more
   than f below the frame's best are left out of every state's sum; default 10.0).  Forward-backward uses its own (minFrwdP, HFB.c:1011). */
int  htkamd_model_set_tm_beam(htkamd_model *m, float tmBeam);
/* The reference's own arithmetic where this library deliberately computes something else (htkamd_model_desc, "WHERE THIS LIBRARY
   DELIBERATELY DIFFERS").  HTKAMD_COMPAT_STREAM_REVISIT: forward-backward on PLAINHS / SHAREDHS sets with NS != 3 streams gives a chain
   state whose tied state was met before in the same Setotprob call -- by a model further right in the beam, or an earlier state of the
   same model -- the value HFB.c:1059 gives it: the float sum of the streams' "sum of the others", halved, i.e. (NS - 1) / 2 times the
   state's log probability.  The recursions, occupation and transition counts see that value, the mixture statistics the first visit's
   "sum of the others", as in the reference (HFB.c:1062-1064,1611).  Batches then run on the general workgroup-per-utterance kernels
   (call before htkamd_fb_prepare).  No effect on sets with one or three streams or on tied-mixture sets (the reference's defect is not
   reached there).
   HTKAMD_COMPAT_SHARED_LOGWT: PLAINHS / SHAREDHS sets whose mixtures share pdfs (~m macros): ConvLogWt passes over a component whose pdf
   it has met before (GoNextMix, HUtil.c:371-394,474-485), so only the first state's weight of a shared pdf becomes a logarithm and the
   others are read as log weights as they stand.  "First" is in HMM scan order (htkamd_model_set_scan_order; either call order works).
   The scoring tables (and every refresh of them after an update) then hold the linear weight where the reference reads it as a log. */
#define HTKAMD_COMPAT_STREAM_REVISIT 1
#define HTKAMD_COMPAT_SHARED_LOGWT   2
int  htkamd_model_set_compat(htkamd_model *m, int flags);
void htkamd_model_destroy(htkamd_model *m);
/* Replace the parameters after a re-estimation pass (same topology). Any pointer may be NULL = unchanged. */
int  htkamd_model_set_params(htkamd_model *m, const float *mean, const float *var, const float *gconst,
                             const float *compWeight, const float *transP);
/* Prepared tables as an HTKLib front-end holds them after ConvDiagC (HUtil.c:413), FixGConsts (HModel.c:5688) and ConvLogWt (HUtil.c:474):
   1/variance [G*D], gConst [G], log mixture weights [C] (LZERO below MINMIX).  The kernels then read exactly these floats (the
   HTKLib shim, shim/htklib_hfb_shim.c, hands over the values of the HMMSet it was given).  Any pointer may be NULL = unchanged. */
int  htkamd_model_set_prepared(htkamd_model *m, const float *ivar, const float *gconst, const float *compLogWt);
/* Read back the prepared device-side values (test/debug aid): each may be NULL. */
int  htkamd_model_get_prepared(htkamd_model *m, float *ivar /*[G*D]*/, float *gconst /*[G]*/,
                               float *compLogWt /*[C]*/, int *minDur /*[nT]*/);

/* ------------------------------------------------------------------------------------------
 * Model definition files: replaces LoadHMMSet (HModel.c:3809) = MakeHMMSet (:3580) + LoadMacroFiles (:3721) with the
 * -d directory search, and SaveHMMSet (:4979) / SaveInOneFile (:4858), for text definitions of one-stream DIAGC
 * continuous-density sets, text or binary (macros ~o ~s ~t ~m ~h ~v"varFloor"; ~u/~v sharing inside a pdf, streams, durations and
 * transforms are rejected with HTKAMD_EMODEL).  Pure host code.
 *   mmf_read    : one master macro file, or one HMM file (a definition without ~h takes `defName` / the file's base name)
 *   mmf_finish  : HMM list "logical [physical]" (NULL: every defined model is its own logical name); physical models still
 *                 undefined are read from dir/name[.ext]; builds the flat description for htkamd_model_create.
 *                 Transition values are logs (GetTransMat, HModel.c:1965), variances as in the file.
 *   mmf_write   : text output of the given parameter arrays (layout of the description; transP logs), into one file
 *                 (macros in the reference's hash-table order) or one file per physical HMM under dir (SAVEGLOBOPTS form).
 * ------------------------------------------------------------------------------------------ */
typedef struct htkamd_mmf htkamd_mmf;
int  htkamd_mmf_create(htkamd_mmf **out);
/* Mixture splitting as HHEd's MU command does it (MixUpCommand HHEd.c:4020, UpMix :2149, SplitMix :1318, HeaviestMix :2112): every
   state with stateSel[state] != 0 (NULL = all) goes to `target` components, or gains -target if target < 0; afterwards
   htkamd_mmf_desc / htkamd_mmf_write reflect the new set.  The step between single-Gaussian and mixture systems in a recipe. */
int  htkamd_mmf_mixup(htkamd_mmf *s, int target, const unsigned char *stateSel);
/* HCompV's PutVFloor (HCompV.c:359-389): "~v varFloor1 <Variance> D" with scale*var, written like WriteVector(" %e"). */
int  htkamd_mmf_write_vfloors(const char *path, const float *var, int D, float scale);
void htkamd_mmf_destroy(htkamd_mmf *s);
int  htkamd_mmf_read(htkamd_mmf *s, const char *path, const char *defName);
int  htkamd_mmf_finish(htkamd_mmf *s, const char *hmmList, const char *dir, const char *ext);
const htkamd_model_desc *htkamd_mmf_desc(const htkamd_mmf *s);
int  htkamd_mmf_num_logical(const htkamd_mmf *s);
const char *htkamd_mmf_logical_name(const htkamd_mmf *s, int i);
int  htkamd_mmf_logical_phys(const htkamd_mmf *s, int i);
int  htkamd_mmf_find_logical(const htkamd_mmf *s, const char *name);     /* physical index or -1 */
const char *htkamd_mmf_phys_name(const htkamd_mmf *s, int h);
const char *htkamd_mmf_parm_kind(const htkamd_mmf *s);                   /* e.g. "MFCC_E_D" */
const float *htkamd_mmf_var_floor(const htkamd_mmf *s);                  /* ~v "varFloor1" [vecSize] or NULL */
int  htkamd_mmf_sharing(const htkamd_mmf *s, int *meanShare /*[numGauss]*/, int *varShare /*[numGauss]*/);   /* ~u / ~v macros; returns the number of Gaussians sharing a vector */
int  htkamd_mmf_write(const htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                      const float *transP, const char *oneFile, const char *dir);
/* the same in HTK's binary form (SaveHMMSet with binary = TRUE, HERest/HHEd -B); htkamd_mmf_read takes either form */
int  htkamd_mmf_write_binary(const htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                             const float *transP, const char *oneFile, const char *dir);
/* SaveHMMSet for a set loaded from several master files (HERest -H macros -H hmmdefs): every macro goes back to the file it was loaded
   from (HModel.c:4388-4470).  masterOut[k] = path for the k-th file read (order of the htkamd_mmf_read calls); models read from files of
   their own beyond those go to dir/<name>. */
int  htkamd_mmf_write_sources(const htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                              const float *transP, const char *const *masterOut, int nMaster, const char *dir, int binary);

/* Script files (-S scp): white-space separated or quoted words (ScriptWord HShell.c:661), each a data file name or an extended
 * file name logical=physical[start,end] (RegisterExtFileName HShell.c:86: frames start..end of `physical`, known as `logical`). */
typedef struct htkamd_scp htkamd_scp;
int  htkamd_scp_read(const char *path, htkamd_scp **out);
void htkamd_scp_free(htkamd_scp *s);
int  htkamd_scp_count(const htkamd_scp *s);
const char *htkamd_scp_logical(const htkamd_scp *s, int i);
const char *htkamd_scp_physical(const htkamd_scp *s, int i);
long htkamd_scp_start(const htkamd_scp *s, int i);      /* -1 = whole file */
long htkamd_scp_end(const htkamd_scp *s, int i);

/* Label OUTPUT: a transcription (one label list, up to two auxiliary labels per entry, as TranscriptionFromLattice HRec.c:2176
 * builds them: word level none; HVite -m [word]; -f -m [model, word]), HVite's -o formatting (FormatTranscription HRec.c:2368)
 * and the writers (SaveHTKLabels HLabel.c:1481: a score column appears only if some label of the list has a non-zero score in it;
 * master label files as HVite -i writes them).  Times in 100 ns units, -1 = absent. */
#define HTKAMD_OUT_NOSCORES   1      /* -o S */
#define HTKAMD_OUT_NOWORDS    2      /* -o W */
#define HTKAMD_OUT_NOTIMES    4      /* -o T */
#define HTKAMD_OUT_NORMSCORES 8      /* -o N : scores per frame */
#define HTKAMD_OUT_TRISTRIP   16     /* -o X : a-b+c -> b */
#define HTKAMD_OUT_CENTRE     32     /* -o C */
#define HTKAMD_OUT_NOMODELS   64     /* -o M */
typedef struct htkamd_trans htkamd_trans;
typedef struct htkamd_mlf_out htkamd_mlf_out;
int  htkamd_trans_create(int maxAux, htkamd_trans **out);
void htkamd_trans_free(htkamd_trans *t);
int  htkamd_trans_add(htkamd_trans *t, double start, double end, const char *name, float score,
                      const char *aux1, float aux1Score, const char *aux2, float aux2Score);
int  htkamd_trans_format(htkamd_trans *t, double frameDur, int states, int models, int flags);
int  htkamd_trans_append_alternative(htkamd_trans *t, htkamd_trans *alt);   /* N-best: further label lists, written after a line of "///"; t owns alt */
int  htkamd_trans_write(const htkamd_trans *t, const char *path);
int  htkamd_mlf_out_open(const char *path, htkamd_mlf_out **out);
int  htkamd_mlf_out_add(htkamd_mlf_out *o, const char *labFile, const htkamd_trans *t);
void htkamd_mlf_out_close(htkamd_mlf_out *o);

/* ------------------------------------------------------------------------------------------
 * Transcriptions: HTK label files (LoadHTKLabels, HLabel.c:748) and master label files with immediate definitions
 * (LoadMasterFile, HLabel.c:1410).  "[start [end]] name [score] ..." per line, times in 100 ns; only the first
 * alternative ("///") is kept; start/end are -1 when absent.  Pure host code.
 * ------------------------------------------------------------------------------------------ */
typedef struct htkamd_labels htkamd_labels;
typedef struct htkamd_mlf htkamd_mlf;
int  htkamd_labels_read(const char *path, htkamd_labels **out);
void htkamd_labels_free(htkamd_labels *l);
int  htkamd_labels_count(const htkamd_labels *l);
const char *htkamd_labels_name(const htkamd_labels *l, int i);
long long htkamd_labels_start(const htkamd_labels *l, int i);
long long htkamd_labels_end(const htkamd_labels *l, int i);
float htkamd_labels_score(const htkamd_labels *l, int i);
int  htkamd_mlf_read(const char *path, htkamd_mlf **out);
void htkamd_mlf_free(htkamd_mlf *m);
const htkamd_labels *htkamd_mlf_find(const htkamd_mlf *m, const char *labFile);   /* first matching pattern, or NULL */

/* ------------------------------------------------------------------------------------------
 * GMM scoring: replaces the state output-probability calls
 *     LogFloat OutP(Observation*,HLink,int)  HModel.h:560 / POutP :561 / SOutP :567 /
 *     MOutP, IDOutP :577-578, and the per-tool caching wrappers ShStrP (HFB.c:898) and
 *     cSOutP/cPOutP (HRec.c:438,512)
 * with one batched call: scores of `ns` tied states for T frames.
 *   dX      device [T*D] row-major feature matrix (Observation.fv[1][1..D] per frame)
 *   dStates device [ns] tied-state indices
 *   dOut    device [ns*ldo]: dOut[k*ldo + t] = log b_{states[k]}(x_t),  ldo >= T
 * Arithmetic = ShStrP/cSOutP: float Mahalanobis sum in dimension order, mixture log-sum with the
 * double-precision LAdd (HMath.c:1576) re-rounded to float after every component: results are
 * bit-identical to the reference.
 * Asynchronous on `stream` (dOut is ready once the stream reaches this point); the call itself allocates and frees nothing after
 * the first few calls.
 * ------------------------------------------------------------------------------------------ */
int htkamd_outp_block(htkamd_model *m, const float *dX, int T, const int *dStates, int ns,
                      float *dOut, int ldo, void *stream);
/* The same scores with a choice of arithmetic:
 *   HTKAMD_SCORE_EXACT  as above (packed-FP32 vector path).
 *   HTKAMD_SCORE_MFMA   Mahalanobis contraction as an fp32 GEMM on the matrix cores ([x^2|x] times per-Gaussian
 *                       coefficients) and a float log-sum-exp over the mixture: within ~1e-4 absolute of the
 *                       reference's float sum (vector sizes up to 40; HTKAMD_EMODEL beyond).  For HERest-style
 *                       accumulation, where the bar is 1e-4 relative on the re-estimated parameters. */
#define HTKAMD_SCORE_EXACT 0
#define HTKAMD_SCORE_MFMA  1
/* Forward-backward only (htkamd_fb_config.scoreMode, may be or-ed with HTKAMD_SCORE_MFMA): the log-add of the alpha/beta
 * recursions (LAdd HMath.c:1576) and the occupation exponentials on fp32 hardware transcendentals; running values stay double.
 * Increment error ~1e-7 absolute: utterance log-probabilities ~1e-9 relative, occupancies / accumulators ~1e-5 relative --
 * inside the 1e-4 bar of the HERest path.  Chains that need the general kernels (models of > 5 states) ignore the bit. */
#define HTKAMD_SCORE_FASTLADD 2
/* The same expanded form on the BF16 matrix pipe, both operands split exactly into three bf16 pieces and the six piece products that
 * matter at fp32 accuracy summed in fp32 (gmm_bf16.hip): the tolerance class of HTKAMD_SCORE_MFMA at ~2.5x its speed (vector sizes up
 * to 48; HTKAMD_EMODEL beyond).  Takes precedence over HTKAMD_SCORE_MFMA when both bits are set. */
#define HTKAMD_SCORE_BF16  4
/* Exact arithmetic with SOutP's rounding (HModel.c:5538-5552: the mixture log-sum kept in double, one rounding to float at the end)
 * instead of ShStrP's / cSOutP's float after every component: what OutP / POutP / SOutP return to a direct caller (HRest, HInit).
 * Bit-identical to the reference's SOutP. */
#define HTKAMD_SCORE_SOUTP 8
/* htkamd_outp_block_mode only, may be or-ed with HTKAMD_SCORE_SOUTP: DOutP's form for DIAGC sets (HModel.c:5347: xmm*xmm/var, the float
 * division) -- what MOutP dispatches to when the set has not been through ConvDiagC, as in HRest / HInit.  Bit-identical to DOutP. */
#define HTKAMD_SCORE_DIAGC 16
/* The expanded form on the FP16 matrix pipe, both operands split into TWO fp16 pieces (11 + 11 significant bits and the signs: fp32's 24)
 * and the three piece products that matter summed in fp32 (gmm_f16.hip): half the matrix instructions of HTKAMD_SCORE_BF16 for the same
 * tolerance class (measured |score - exact| rms 3.4e-5 against 3.2e-5), ~1.45x its speed.  fp16's range is bridged by a power-of-two
 * scale per term, chosen from the model on the device; a feature value far outside anything the model describes (or a model whose
 * coefficients span more than the format) is DETECTED, not computed wrongly: htkamd_fb_results / htkamd_outp_block_mode return
 * HTKAMD_ERANGE and the pass is to be repeated with HTKAMD_SCORE_BF16.  Forward-backward and htkamd_outp_block_mode (vector sizes up
 * to 45); takes precedence over HTKAMD_SCORE_BF16 and HTKAMD_SCORE_MFMA. */
#define HTKAMD_SCORE_F16   32
#define HTKAMD_SCORE_FAST  (HTKAMD_SCORE_MFMA | HTKAMD_SCORE_FASTLADD)
#define HTKAMD_SCORE_FASTEST (HTKAMD_SCORE_F16 | HTKAMD_SCORE_FASTLADD)
int htkamd_outp_block_mode(htkamd_model *m, const float *dX, int T, const int *dStates, int ns,
                           float *dOut, int ldo, int scoreMode, void *stream);
/* HTKAMD_SCORE_F16 through htkamd_outp_block_mode: the calls are asynchronous, so the range flag they raise is the model's.  This waits
 * for the stream and returns HTKAMD_ERANGE if a call since the last check raised it (and clears it): the scores of those calls are to
 * be recomputed with HTKAMD_SCORE_BF16.  (Forward-backward passes carry their own flag: htkamd_fb_results.) */
int htkamd_model_f16_check(htkamd_model *m, void *stream);

/* ------------------------------------------------------------------------------------------
 * Baum-Welch accumulators: MuAcc / VaAcc / WtAcc / TrAcc of HTrain.h:211-232 plus the
 * per-model example counters (hmm->hook, HFB.c:1768-1772) and HERest's totals
 * (totalPr, totalT: HERest.c:779-780), kept on the device as ONE flat vector of doubles so
 * that the parallel-mode merge (HERest -p N dump + -p 0 load: HTrain.c:1453,1625) is a single
 * sum all-reduce over that vector.
 * Vector layout (offsets via htkamd_accs_layout):
 *   mu[G*D] muOcc[G] va[G*D] vaOcc[G] wt[C] wtOcc[S] tr[transOff[nT]] trOcc[sum N] nEgs[H]
 *   totalPr totalT nUttDone nUttSkipped nEval
 * ------------------------------------------------------------------------------------------ */
typedef struct htkamd_accs htkamd_accs;
typedef struct {
   size_t mu, muOcc, va, vaOcc, wt, wtOcc, tr, trOcc, nEgs, totalPr, totalT, nUttDone, nUttSkipped, nEval;
   size_t total;              /* number of doubles */
} htkamd_accs_layout;

int  htkamd_accs_create(htkamd_model *m, htkamd_accs **out);
void htkamd_accs_destroy(htkamd_accs *a);
int  htkamd_accs_zero(htkamd_accs *a, void *stream);                          /* ZeroAccs HTrain.c */
int  htkamd_accs_get_layout(const htkamd_accs *a, htkamd_accs_layout *out);
int  htkamd_accs_device_vector(htkamd_accs *a, double **dVec, size_t *n);      /* for the all-reduce */
int  htkamd_accs_download(htkamd_accs *a, double *hostVec /*[layout.total]*/, void *stream);
int  htkamd_accs_upload_add(htkamd_accs *a, const double *hostVec, void *stream); /* LoadAccs: adds */

/* The exchange step of a multi-GPU pass from C: sum all-reduce of the accumulator vector over RCCL (xGMI inside a node), one process
 * per GPU.  Replaces the dump / load round trip of HERest's parallel mode (DumpAccs HTrain.c:1453 per `-p k` process, LoadAccs
 * HTrain.c:1625 + sum in the `-p 0` process, HERest.c:514-550); afterwards every rank holds the same sums and runs the same update.
 *   comm_unique_id : rank 0 obtains the 128-byte rendezvous id and passes it to the other ranks (file, socket, environment ...)
 *   comm_init      : every rank, same id; nRanks == 1 needs neither RCCL nor an id
 *   accs_allreduce : in place, asynchronous on `stream`
 * RCCL is bound at run time (librccl.so.1); HTKAMD_ENODEV if it cannot be found when nRanks > 1. */
typedef struct htkamd_comm htkamd_comm;
int  htkamd_comm_unique_id(void *id128);
int  htkamd_comm_init(htkamd_comm **out, int nRanks, int rank, const void *id128);
void htkamd_comm_destroy(htkamd_comm *c);
int  htkamd_comm_ranks(const htkamd_comm *c);
int  htkamd_accs_allreduce(htkamd_accs *a, htkamd_comm *c, void *stream);
/* The exchange with fp32 on the wire (half the bytes: SURVEY §8(e)'s 25.9 MB): every rank rounds its fp64 partial statistics to float
 * once, RCCL sums floats, the sums return to the fp64 vector; nEgs / totalPr / totalT / the counters stay fp64 (a second, small
 * all-reduce).  The reference's merge adds float dumps (LoadAccs HTrain.c:1625-1687): this is tighter.  accs_wire_round applies the
 * rounding alone (one rank's share of it, for emulating the exchange in tests); comm_agree_max = max over the ranks of one int,
 * synchronous -- a decision all ranks must take alike (the fp16 -> bf16 fallback of an iteration). */
#define HTKAMD_WIRE_F64 0
#define HTKAMD_WIRE_F32 1
int  htkamd_accs_allreduce_wire(htkamd_accs *a, htkamd_comm *c, int wire, void *stream);
int  htkamd_accs_wire_round(htkamd_accs *a, void *stream);
int  htkamd_comm_agree_max(htkamd_comm *c, int *value, void *stream);
/* The exchange in parts, behind a pass in two phases (htkamd_fb_execute_begin / _mix): the statistics of tied states [state0, state1) travel
 * while the next range of states is still being summed.  One logical exchange per iteration (what `HERest -p 0 HER*.acc` merges,
 * HERest.c:514-557, HTrain.c:1625-1687), cut along the accumulator vector.
 *   htkamd_accs_state_ranges   the ranges of the vector that belong to those states -- mu, muOcc, va, vaOcc of their Gaussians, wt of their
 *                              components, wtOcc -- and with `withRest` what no state owns and the whole-vector exchange sends as
 *                              statistics (tr, trOcc): up to 7 ranges in off[] / len[] (doubles), their number in *n.  Needs a set whose
 *                              components own their Gaussians in state order (compGauss[c] == c: no shared pdfs); else HTKAMD_EMODEL,
 *                              and the caller exchanges the vector whole.
 *   htkamd_accs_pack_ranges    the ranges, one behind the other, into `dst` as floats (HTKAMD_WIRE_F32) or doubles; _unpack_ranges back.
 *                              For hosts with a collective of their own (bench.py: torch.distributed).
 *   htkamd_accs_allreduce_states   pack, sum over the ranks (RCCL), unpack -- all on `stream`; `withRest` adds tr / trOcc to the part and
 *                              sums the counters behind them (nEgs ... nEval) in fp64 as htkamd_accs_allreduce_wire does.  Calls with
 *                              withRest = 1 once per iteration, and every state once, leave every rank with the vector the whole
 *                              exchange would have left (bit for bit on the fp64 wire, for any ring order on the fp32 wire: the same
 *                              values are rounded and summed). */
int  htkamd_accs_state_ranges(htkamd_accs *a, int state0, int state1, int withRest, size_t off[7], size_t len[7], int *n);
int  htkamd_accs_pack_ranges(htkamd_accs *a, int n, const size_t *off, const size_t *len, int wire, void *dst, void *stream);
int  htkamd_accs_unpack_ranges(htkamd_accs *a, int n, const size_t *off, const size_t *len, int wire, const void *src, void *stream);
int  htkamd_accs_allreduce_states(htkamd_accs *a, htkamd_comm *c, int wire, int state0, int state1, int withRest, void *stream);

/* HTK parameter files: header + big-endian float rows, _C compression and _K checksum on input
 * (ReadHTKHeader HWave.c:1408, OpenParmChannel HParm.c:3561, GetParm :3464, UpdateCRCC :3357).  *data is malloc'd
 * (release with htkamd_free); *kind comes back without the _C/_K bits. */
int  htkamd_parm_read(const char *path, float **data, int *nFrames, int *nCols, int *sampPeriod, int *kind);
int  htkamd_parm_write(const char *path, const float *data, int nFrames, int nCols, int sampPeriod, int kind, int withCrc);
void htkamd_free(void *p);
/* Waveform files, the sources of the MFCC front end: SOURCEFORMAT = WAV (RIFF 16-bit mono PCM, GetWAVHeaderInfo HWave.c:1052) or
 * HTK (big-endian WAVEFORM file).  *samples is malloc'd (htkamd_free); *sampPeriod in 100 ns units (625 at 16 kHz). */
#define HTKAMD_WAVE_HTK 1
#define HTKAMD_WAVE_WAV 2
int  htkamd_wave_read(const char *path, int format, short **samples, long *nSamples, double *sampPeriod);

/* Accumulator files (HERN.acc) for exchanging statistics with the reference's parallel mode: DumpAccs
 * (HTrain.c:1453) + trailer (HERest.c:546-548) and LoadAccs (HTrain.c:1625; adds).  Pure host functions on the
 * flat vector; `names[h]` is the physical HMM name in definition order (the HMM list).  Files follow the
 * reference's HMM-scan order, so `HERest -p 0` reads what is written here and vice versa. */
int htkamd_accs_layout_from_desc(const htkamd_model_desc *d, htkamd_accs_layout *out);
int htkamd_hmm_scan_order(const char *const *names, int H, int *order);
int htkamd_accs_dump_file(const htkamd_model_desc *d, const double *hostVec, const char *const *names, int uFlags, const char *path);
int htkamd_accs_load_file(const htkamd_model_desc *d, double *hostVec, const char *const *names, int uFlags, const char *path);
/* The same for a set with tied mean / variance vectors (meanShare / varShare as in htkamd_model_set_sharing, either may be NULL): a
 * shared vector has ONE MuAcc / VaAcc record in the file, where the scan first meets it (HTrain.c:1484-1493).  The writer puts the sum
 * over the sharers there; the reader adds the record to the vector's first sharer, which is all htkamd_model_update needs. */
int htkamd_accs_dump_file_shared(const htkamd_model_desc *d, const double *hostVec, const char *const *names, int uFlags,
                                 const int *meanShare, const int *varShare, const char *path);
int htkamd_accs_load_file_shared(const htkamd_model_desc *d, double *hostVec, const char *const *names, int uFlags,
                                 const int *meanShare, const int *varShare, const char *path);
/* HERest -s file: per physical HMM (scan order) its index, quoted name, example count and the occupation of each emitting
 * state -- StatReport / PrintStats (HERest.c:708-747), the input of HHEd's RO / TB / TC commands. */
int htkamd_stats_write_file(const htkamd_model_desc *d, const double *hostVec, const char *const *names, const char *path);

/* ------------------------------------------------------------------------------------------
 * Model update after a pass: UpdateModels (HERest.c:1326) -> MLUpdateModels (HERest.c:1262) with
 * UpdateTrans :795, UpdateWeights :897 (+FloorMixes :819), UpdateVars :1045, UpdateMeans :974 and
 * FixGConsts (HModel.c:5688).  Runs on the host (milliseconds, as in the reference) from a host copy of
 * the summed accumulator vector, then refreshes the device tables.  With HTKAMD_UPMAP in uFlags: MAPUpdateModels (HMap.c:413)
 * with its UpdateWeights :205, UpdateVars :314, UpdateMeans :279 -- host update only.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
   int   minEgs;            /* HERest -m, default 3 (HERest.c:96)                                  */
   float minVar;            /* HERest -v, default 0.0 (HERest.c:95)                                */
   float mixWeightFloor;    /* HERest -w f gives f*MINMIX (HERest.c:425), default 0.0              */
   int   uFlags;            /* HTKAMD_UP* bits                                                     */
   int   singleProcess;     /* 1 = parMode -1: ForceDiagC/ConvExpWt round trips (HERest.c:1336-1339) */
   const float *varFloor;   /* [vecSize] per-component floor = the ~v "varFloor1" macro (SetVFloor HModel.c:3512: when the
                               macro exists it replaces minVar); NULL = minVar everywhere                */
   int   rowNormalise;      /* 1 = transition rows renormalised by their sum, as the isolated-unit trainer does
                               (RestTransP HRest.c:1015); 0 = HERest's UpdateTrans                       */
   float mapTau;            /* HTKAMD_UPMAP only: HMAP: MAPTAU, the weight of the prior (HMap.c:82, default there 20.0).  With UPMAP the
                               other fields mean what HMap's own configuration means: minEgs = HMAP: MINEGS (default 0), minVar = HMAP:
                               MINVAR (default 0.0), mixWeightFloor = MINMIX * HMAP: MIXWEIGHTFLOOR; HTKAMD_UPTRANS is refused
                               (HError 999 "No support for MAP updating transition probabilities", HMap.c:434)   */
   float mapMinObs;         /* HMAP: MINOBS: a mean counts as observed in stats.nMapObserved above this occupation          */
} htkamd_update_config;
typedef struct {
   int nFloorVar, nFloorVarMix;      /* "Total %d floored variance elements in %d different mixes"  */
   int nSkippedHmm;                  /* models copied because they had < minEgs examples (-2331)     */
   int nNoTransOut, nNoMixUse, nNoVarUse;   /* warnings -2326 / -2330                                */
   int nWeightAboveOne;              /* re-estimated mixture weights above 1.001: fatal HError 2393 in UpdateWeights (HERest.c:926);
                                        the update is carried out (weights clamped to 1) and the call returns HTKAMD_EMODEL */
   int nMapObserved;                 /* HTKAMD_UPMAP: "Observed components (means) %d of ..." (HMap.c:452)                    */
} htkamd_update_stats;
int htkamd_model_update(htkamd_model *m, const htkamd_accs *accs, const double *hostVec,
                        const htkamd_update_config *cfg, htkamd_update_stats *stats);
/* The same update on the device, straight from the accumulator vector where it lies (after the all-reduce): nothing but the
   transition matrices (for the host's minimum-duration table) and the counters of `stats` crosses PCIe, and every table the kernels
   read -- 1/variance, the interleaved scoring rows, log weights, the matrix-core fragment table -- is rebuilt in place.
   Arithmetic as htkamd_model_update; `stats` may be NULL.  Synchronises `stream` before returning. */
int htkamd_model_update_device(htkamd_model *m, htkamd_accs *accs, const htkamd_update_config *cfg, htkamd_update_stats *stats, void *stream);
/* The same in two halves for a host loop that wants the next pass queued behind the update's kernels before it waits: _begin launches
   everything and the copy of the transition matrices + counters and returns; _end waits for that copy only (an event, not the stream) and
   folds it into the host tables.  Between the two the host tables are the OLD ones: a pass queued in between runs on the new parameters
   (stream order) with batch tables prepared for the old minimum durations -- after _end, htkamd_fb_prepared_current tells whether that
   pass has to be repeated (a minimum duration changed: rare).  One update in flight per model. */
int htkamd_model_update_device_begin(htkamd_model *m, htkamd_accs *accs, const htkamd_update_config *cfg, void *stream);
int htkamd_model_update_device_end(htkamd_model *m, htkamd_update_stats *stats);
/* Current parameters (DIAGC variances, linear weights, log transitions); any pointer may be NULL. */
int htkamd_model_get_params(htkamd_model *m, float *mean, float *var, float *gconst, float *compWeight, float *transP);

/* ------------------------------------------------------------------------------------------
 * Forward-backward over a batch of utterances: replaces, per utterance,
 *     Boolean FBFile(FBInfo*, UttInfo*, char *datafn)          HFB.h:143 / HFB.c:1923
 * i.e. StepBack (SetBeamTaper, Setotprob, SetBeta with the beta beam and its retry loop) and
 * StepForward (StepAlpha with the alpha beam, SetOcct, UpMixParms, UpTranParms).
 * The label sequence of an utterance is the list of physical-HMM indices CreateInsts
 * (HFB.c:508) derives from the transcription.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
   double pruneInit, pruneInc, pruneLim;   /* HERest -t f [i l]; HTKAMD_NOPRUNE = off (HERest.c:121-123) */
   float  minFrwdP;                        /* HFB MINFORPROB / HERest -c, default 10.0 (HFB.c:83)        */
   int    uFlags;                          /* HTKAMD_UP* bits                                            */
   int    scoreMode;                       /* HTKAMD_SCORE_EXACT (0, default), or HTKAMD_SCORE_MFMA | HTKAMD_SCORE_FASTLADD bits */
} htkamd_fb_config;

typedef struct {
   int          nUtt;
   const float *dX;          /* device [frameOff[nUtt]*D] all utterances' frames, row-major  */
   const int   *frameOff;    /* host [nUtt+1]                                               */
   const int   *labOff;      /* host [nUtt+1]                                               */
   const int   *labs;        /* host [labOff[nUtt]] physical HMM index of each label        */
} htkamd_batch_desc;

/* per-utterance status values */
#define HTKAMD_UTT_OK        1
#define HTKAMD_UTT_SKIPPED   0        /* HError -7324: no path / qt > T (HFB.c:1339-1356)   */
#define HTKAMD_UTT_ETEE     (-7332)   /* tee-model placement (HFB.c:557,564)                */
#define HTKAMD_UTT_EALPHA   (-7390)   /* alpha prune failed (HFB.c:706,718)                 */

typedef struct htkamd_fb htkamd_fb;

int  htkamd_fb_create(htkamd_model *m, htkamd_fb **out);
void htkamd_fb_destroy(htkamd_fb *fb);
/* Test aid; call before prepare.  bit 0: keep every alpha column (T*cells doubles more per utterance);
   bit 1: force the general workgroup-per-utterance kernels even where the wave-per-utterance path applies;
   bit 2: keep utterances off the lane-per-chain-state kernels (they then take the lane-per-model ones). */
int  htkamd_fb_set_debug(htkamd_fb *fb, int on);
/* Host part of CreateInsts/SetBeamTaper for the whole batch + upload of the chain tables.
   Limits per utterance: chains of up to 512 models when no model has more than 5 states, otherwise up to
   1024 model states; an utterance beyond them fails the call with HTKAMD_EINVAL (nothing is truncated). */
int  htkamd_fb_prepare(htkamd_fb *fb, const htkamd_batch_desc *batch, void *stream);
/* The batch tables depend on the transcriptions and on the models' minimum durations only, so a batch may be prepared ahead of the
   pass that uses it (e.g. for the next EM iteration while this one's kernels run) and executed any number of times.  Returns 0 when
   a model update has since changed a minimum duration (a transition reached or left zero): htkamd_fb_execute then refuses the batch
   (HTKAMD_EINVAL) and it must be prepared again. */
int  htkamd_fb_prepared_current(const htkamd_fb *fb);
/* Device part: scores, beta pass, alpha pass + statistics into `accs`. Asynchronous on `stream`. */
int  htkamd_fb_execute(htkamd_fb *fb, const htkamd_fb_config *cfg, htkamd_accs *accs, void *stream);
/* Waits for the stream and copies per-utterance log-probabilities (utt->pr) and status. */
/* The pass in two phases (multi-GPU hosts: a range of states' statistics is exchanged while the next is being summed).  _begin: everything
   but the state-bucketed mixture statistics (UpMixParms HFB.c:1426 for the left-to-right path's surviving pairs); *deferred = 1 when those
   wait for _mix, 0 when the pass was of another kind and is complete.  _mix: the statistics of tied states [state0, state1); every state
   once per pass, any order.  Behind _mix the ranges htkamd_accs_state_ranges names for those states are final on this rank; what no state
   owns is final behind _begin.  htkamd_fb_execute = _begin + _mix(0, numStates). */
int  htkamd_fb_execute_begin(htkamd_fb *fb, const htkamd_fb_config *cfg, htkamd_accs *accs, void *stream, int *deferred);
int  htkamd_fb_execute_mix(htkamd_fb *fb, int state0, int state1, void *stream);
int  htkamd_fb_results(htkamd_fb *fb, double *pr /*[nUtt]*/, int *status /*[nUtt]*/, void *stream);
/* For host loops that queue the next pass before they read this one's results: the copy of the results queued on `stream` (the stream
   htkamd_fb_execute ran on) right behind the pass; htkamd_fb_results then only waits for that copy. */
int  htkamd_fb_results_begin(htkamd_fb *fb, void *stream);
/* Number of (frame, chain-state) output-probability evaluations the reference would perform
   for the prepared batch without pruning (Setotprob visits) -- the unit of the throughput metric. */
long long htkamd_fb_frame_states(const htkamd_fb *fb);
/* Debug/test access to one utterance's trellis after execute (device -> host copies):
   beta/alpha [T*Q*maxN] (NaN where the reference holds no vector), outp [T*Q*maxN], beams [T]. */
int  htkamd_fb_get_trellis(htkamd_fb *fb, int utt, double *beta, double *alpha, float *outp,
                           int *qLo, int *qHi, int *aLo, int *aHi, int *T, int *Q, int *maxN, void *stream);
/* Seconds spent in the kernels of the last execute, measured with HIP events on `stream`:
   out[0]=scoring (the dispatch's own start -> stop, hipExtLaunchKernel: what a kernel trace reports even when other streams share
   the device) out[1]=beta out[2]=alpha+occ/trans out[3]=mixture statistics (intervals between stream events). Synchronises.
   An interval the pass did not measure (htkamd_fb_set_event_mode(fb, 1)) reads -1, and so does a sum with one. */
int  htkamd_fb_kernel_times(htkamd_fb *fb, double out[4]);
/* the same with the alpha pass and the left-to-right path's frame-parallel statistics apart: scoring, beta, alpha, statistics, mixture statistics */
int  htkamd_fb_kernel_times5(htkamd_fb *fb, double out[5]);
/* Which events htkamd_fb_execute records for the two calls above: 0 (default) the scoring dispatch's own start / stop plus stream events between
   the kernels (each a barrier packet: 20 - 40 us of a 2 ms pass in all); 1 the scoring dispatch's only -- the other intervals then read -1. */
int  htkamd_fb_set_event_mode(htkamd_fb *fb, int mode);
/* the last pass's mixture statistics in units (UpMixParms HFB.c:1573-1721): out[0] (frame, state) pairs past the MINFORPROB prune, out[1] (frame,
   state, component) triples accumulated; -1 where the pass did not count them (sets of several streams, states of more than 16 components) */
int  htkamd_fb_mix_counts(htkamd_fb *fb, long long out[2]);
/* the prepared batch's work on the matrix cores, counted on the host from its task list (the 32 x 32 x 16 pair kernels, one state per
   16-component tile): out[0] matrix instructions (per operand-piece product and k-step 1; multiply by the kernel's 6 x 5 or 3 x 5) a pass
   issues per (wavefront, pair of states) it does not sit out, out[1] the same with every wavefront of a task working on every pair (before the
   per-state frame ranges of Setotprob, HFB.c:1014, reached the kernel), out[2] what the (frame, chain state) evaluations need: units / 64 */
int  htkamd_fb_score_work(const htkamd_fb *fb, long long out[3]);

/* ------------------------------------------------------------------------------------------
 * Viterbi forced alignment of a batch (HVite -a): replaces, per utterance, the frame loop of
 * HVite.c:640-710 ProcessFile on the alignment network of DoAlignment (HVite.c:830):
 *     StartRecognition / ProcessObservation / CompleteRecognition      HRec.h:170-190
 *     TranscriptionFromLattice (HRec.c:2176) for the state (-f) and model (-m) level labels
 * The network is the linear chain of the transcription's physical models (LatticeFromLabels +
 * ExpandWordNet with one pronunciation per word).  genBeam is HVite's -t value (1e10 = off).
 * Results are per chain state (one segment per emitting state, start = -1 if the state was skipped):
 * frames [segStart, segEnd) 0-based and segScore = acoustic score of the segment, and per model
 * [modStart, modEnd), modScore -- the numbers HVite prints as "start end s<j> score model score".
 * Token likelihoods are the same double additions in the same order as HRec's, on bit-exact output
 * probabilities, so segmentations are bit-identical to the reference's.
 * Chains of <= 512 models of <= 5 states run on 1..8 wavefronts per utterance; longer chains and larger
 * models a workgroup per utterance (same results).  Word networks: htkamd_decode_* below.
 * ------------------------------------------------------------------------------------------ */
typedef struct htkamd_viterbi htkamd_viterbi;
int  htkamd_viterbi_create(htkamd_model *m, htkamd_viterbi **out);
void htkamd_viterbi_destroy(htkamd_viterbi *v);
int  htkamd_viterbi_align(htkamd_viterbi *v, const htkamd_batch_desc *batch, float genBeam, void *stream);
/* The same with the state scores in another arithmetic of the EXACT family: HTKAMD_SCORE_SOUTP (SOutP's double accumulation with one
 * float rounding, HModel.c:5538) and / or HTKAMD_SCORE_DIAGC (DOutP's division by the variance, HModel.c:5347) -- what HInit's
 * ViterbiAlign (HInit.c:792, OutP on a set that has not been through ConvDiagC) computes.  0 = htkamd_viterbi_align. */
int  htkamd_viterbi_align_mode(htkamd_viterbi *v, const htkamd_batch_desc *batch, float genBeam, int scoreMode, void *stream);
int  htkamd_viterbi_sizes(const htkamd_viterbi *v, size_t *nSeg /* sum of chain states */, size_t *nMod /* sum of models */);
int  htkamd_viterbi_results(htkamd_viterbi *v, int *segStart, int *segEnd, double *segScore,
                            int *modStart, int *modEnd, double *modScore,
                            double *total /*[nUtt]*/, int *status /*[nUtt]*/, void *stream);

/* ------------------------------------------------------------------------------------------
 * Recognition network: word lattice (SLF) + dictionary -> model-level network, for context-independent sets.
 * Replaces ReadLattice (HNet.c:631), ReadDict (HDict.c:224) and ExpandWordNet (HNet.c:3438, xc == 0 form): per lattice
 * node and pronunciation a chain of HMM nodes ending in a WORD node, NULL nodes for !NULL words, lattice arcs as links
 * carrying the LM log probability, node 0 = the network's initial null node, node 1 = its final null node
 * (AddInitialFinal HNet.c:2180).  Pure host code.
 * ------------------------------------------------------------------------------------------ */
#define HTKAMD_NODE_HMM  0     /* model = physical HMM index */
#define HTKAMD_NODE_WORD 1     /* model = pronunciation index (htkamd_net_out_sym), pronProb = log pron prob */
#define HTKAMD_NODE_NULL 2
typedef struct {
   int nNodes, nLinks, nProns, initial, final;
   const int   *kind, *model;
   const float *pronProb;
   const int   *linkOff;     /* [nNodes+1] links of node n: linkOff[n]..linkOff[n+1) */
   const int   *linkDest;
   const float *linkLike;    /* LM log probability of the link (NetLink.like) */
} htkamd_net_desc;
typedef struct htkamd_net htkamd_net;
int  htkamd_net_build(const char *slfPath, const char *dictPath, const htkamd_mmf *hmms, htkamd_net **out);
/* The same with HNet's configuration switches (HNet.c:122-127; HVite reads them from its -C file):
 *   ALLOWXWRDEXP   contexts may cross word boundaries: when the model set defines contexts and the dictionary cannot be built from
 *                  word-internal models alone (or one of the FORCE switches is set) the network is expanded with CROSS-WORD contexts
 *                  (ExpandWordNet HNet.c:3438 with xc > 0): a word's first model depends on the last context phone of the word before,
 *                  its last model on the first context phone of the word after; context-free phones (sp) and null words pass contexts on
 *   FORCECXTEXP    expand contexts even when every dictionary phone is a model name      FORCELEFTBI / FORCERIGHTBI   biphone names only
 * flags = 0 is htkamd_net_build.  With cross-word expansion `hmms` must stay alive as long as the network (htkamd_net_seq_models). */
#define HTKAMD_NET_ALLOWXWRDEXP 1
#define HTKAMD_NET_FORCECXTEXP  2
#define HTKAMD_NET_FORCELEFTBI  4
#define HTKAMD_NET_FORCERIGHTBI 8
int  htkamd_net_build_ex(const char *slfPath, const char *dictPath, const htkamd_mmf *hmms, int flags, htkamd_net **out);
int  htkamd_net_is_xwrd(const htkamd_net *n);
/* Physical models of pronunciation `pron` when it stands between the pronunciations prevPron and nextPron (-1: utterance boundary;
 * pronunciations without phones are transparent, so pass the nearest one that has phones): what HVite -m labels for a recognised
 * word sequence.  Without cross-word expansion this is htkamd_net_pron_models.  Returns the number of models (may exceed `max`). */
int  htkamd_net_seq_models(const htkamd_net *n, int pron, int prevPron, int nextPron, int *models, int max);
/* Alignment network of HVite -a from a word-level transcription: LatticeFromLabels (HNet.c:1516: one node per label, the boundary
   word of HVite -b at both ends when non-NULL) + the same expansion; decoding it with htkamd_decoder_* is HVite -a (DoAlignment
   HVite.c:830), word- or model-level (-m) labels. */
int  htkamd_net_build_words(const char *const *words, int nWords, const char *boundary, const char *dictPath,
                            const htkamd_mmf *hmms, htkamd_net **out);
void htkamd_net_destroy(htkamd_net *n);
const htkamd_net_desc *htkamd_net_get(const htkamd_net *n);
const char *htkamd_net_out_sym(const htkamd_net *n, int pron);
/* physical models of pronunciation `pron` after context expansion; returns their number (which may exceed max) */
int  htkamd_net_pron_models(const htkamd_net *n, int pron, int *models, int max);
const char *htkamd_net_word_name(const htkamd_net *n, int pron);

/* ------------------------------------------------------------------------------------------
 * Network decoding of a batch (HVite -w net): replaces, per utterance, InitVRecInfo / StartRecognition /
 * ProcessObservation / CompleteRecognition (HRec.h:150-190) with nToks = 1 and TranscriptionFromLattice (HRec.c:2176)
 * for the word-level 1-best labels.  genBeam / wordBeam = HVite -t / -v (1e10 = off), lmScale -s, wordPen -p, prScale -r.
 * LikeToWord look-ahead (HRec.c:1172) depends on the LM scale, hence lmScale at creation.
 * Results per utterance u: nWords[u] (-1: no token reached the end of the network, -3: more than maxWords words; from the walk of HRec's
 * instance list -- HTKAMD_ORDER_EXACT and N-best runs only, in the default HTKAMD_ORDER_AUTO mode such an utterance keeps the batch kernel's
 * answer --: -4 out of Path records, -5 more list appends in one frame than 8 nNodes + 1024, -6 zero-time nodes nested deeper than 64), and for
 * word w < nWords[u] at [u*maxWords + w]: pronunciation index (htkamd_net_out_sym), frames [start, end), score = LArcTotLike
 * (acoustic + scaled LM + scaled pron prob + word penalty, HNet.h:257), wordLm (may be NULL) = the arc's LM log probability
 * (LArc.lmlike: what HVite -m/-f print, scaled and with the word penalty, as the word's auxiliary score); total[u] = likelihood
 * of the final token.  Model-level labels of the recognised words (HVite -m with -w) = htkamd_viterbi_align on the chain of
 * htkamd_net_pron_models of the recognised pronunciations: for a fixed word sequence the LM terms are constants, so the best
 * alignment of that chain is the decoder's path.
 * Models of up to 8 states; tee models may not have more than 95 predecessors.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
   float genBeam, wordBeam, lmScale, wordPen, prScale;
   int   scoreMode;   /* HTKAMD_SCORE_EXACT (0): scores, token likelihoods and therefore paths are the reference's bit for bit;
                         HTKAMD_SCORE_MFMA: matrix-core scores (1e-4 class) -- same words unless two paths tie within that */
   int   maxActive;   /* HVite -u: maximum-model pruning (ProcessObservation HRec.c:1966-1985); 0 = off */
} htkamd_decode_config;
typedef struct htkamd_decoder htkamd_decoder;
int  htkamd_decoder_create(htkamd_model *m, const htkamd_net_desc *net, float lmScale, htkamd_decoder **out);
/* N-best token passing and lattice generation (HVite -n N [-z ext]; HRec.c with nToks > 1: TokSetMerge :279, StepWord2's NxtPath
 * chains :1046, CreateLattice :1679 / LatFromPaths :1512).  Every state keeps its best token and up to nToks-1 alternatives that
 * differ in the word they came from; the lattice of an utterance = the word ends reachable from the final token set:
 *   node 0 = start, node 1 = end (both !NULL, nodePron -1), the others word ends {frame, pronunciation, likelihood of the Path};
 *   arc = one way into a word end: aclike, lmlike (unscaled LM log probability), prlike (log pronunciation probability).
 * nToks: 2..8;  nBeam: alternatives more than this below the frame's best token are dropped (HVite uses its -t value);
 * cfg->maxActive must be 0.  Outputs per utterance u at [u*maxLatNodes + i] / [u*maxLatArcs + j]; nNodes[u] = -1: no token reached
 * the end, -3: the lattice did not fit.  Pointers other than nNodes / nArcs may be NULL. */
typedef struct {
   int *nNodes, *nArcs;
   int *nodeFrame, *nodePron, *nodeNet;
   double *nodeLike;
   int *arcStart, *arcEnd;
   float *arcAc, *arcLm, *arcPr;
   double *arcScore;
   double *total;
} htkamd_lattice_out;
int  htkamd_decoder_run_lattice(htkamd_decoder *d, const htkamd_decode_config *cfg, int nToks, float nBeam, const float *dX, const int *frameOff, int nUtt,
                                int maxLatNodes, int maxLatArcs, const htkamd_lattice_out *out, void *stream);
/* The same with alignment records inside the arcs (HVite -n together with -m: alignMode & 1, model records / -f: alignMode & 2, state
 * records; LatFromPaths' lAlign, HRec.c:1582-1656, as the reference built with -DPHNALG -- HTKLib/Makefile.in:45 -- writes it: the records
 * of a relative token carry the BEST token's likelihoods, so an alternative's alignment likelihoods need not add up to its arc's).  Arc j
 * of utterance u owns records [arcAlignOff[u*(maxLatArcs+1) + j], .. + j + 1) of alState (-1: a model record) / alModel (physical model)
 * / alDur (frames) / alLike at u*maxAlign + i.  Alignment records are kept for the whole utterance on the device (frames x models x
 * tokens of them): meant for rescoring-size networks.  nNodes[u] = -3 also when they (or maxAlign) do not fit. */
typedef struct {
   int *arcAlignOff;
   int *alState, *alModel, *alDur;
   float *alLike;
} htkamd_lattice_align_out;
int  htkamd_decoder_run_lattice_align(htkamd_decoder *d, const htkamd_decode_config *cfg, int nToks, float nBeam, int alignMode, const float *dX, const int *frameOff,
                                      int nUtt, int maxLatNodes, int maxLatArcs, int maxAlign, const htkamd_lattice_out *out, const htkamd_lattice_align_out *alOut,
                                      void *stream);
/* One utterance's lattice for the host-side functions below (pointers into the arrays above). */
typedef struct {
   int nNodes, nArcs;
   const int *nodeFrame, *nodePron;
   const double *nodeLike;
   const int *arcStart, *arcEnd;
   const float *arcAc, *arcLm, *arcPr;
   float lmScale, wordPen, prScale;     /* of the recognition run (header fields, LArcTotLike) */
   double frameDur;                     /* seconds per frame */
} htkamd_lattice;
/* WriteLattice (HNet.c:631): the SLF text file, byte for byte the reference's for the supported fields.  format = HVite -q letters as
 * bits; 0 = HTKAMD_LAT_DEFAULT (t v a l).  utterance / lmName / vocabName: header lines (NULL = left out). */
#define HTKAMD_LAT_ALABS  0x0001
#define HTKAMD_LAT_LBIN   0x0002
#define HTKAMD_LAT_TIMES  0x0008
#define HTKAMD_LAT_PRON   0x0010
#define HTKAMD_LAT_ACLIKE 0x0020
#define HTKAMD_LAT_LMLIKE 0x0040
#define HTKAMD_LAT_ALIGN  0x0080
#define HTKAMD_LAT_PRLIKE 0x0400
#define HTKAMD_LAT_DEFAULT (HTKAMD_LAT_TIMES | HTKAMD_LAT_PRON | HTKAMD_LAT_ACLIKE | HTKAMD_LAT_LMLIKE)
int  htkamd_lattice_write(const htkamd_lattice *lat, const htkamd_net *net, const char *path, const char *utterance, const char *lmName,
                          const char *vocabName, int format);
/* ... with the arcs' alignment records (htkamd_decoder_run_lattice_align) as WriteLattice's OutputAlign writes them (HNet.c:503-516):
 * `d=:label[,duration][,likelihood]: ...` per arc that has records -- label = the physical model's name for a model record, "s<j>" for a
 * state record when model records were made too (models != 0), "<model>[<j>]" otherwise; format bits HTKAMD_LAT_ALIGN / _ALDUR / _ALLIKE
 * (HVite's default -q is all of t v a l d with durations and likelihoods: HTKAMD_LAT_DEFAULT_ALIGN). */
typedef struct {
   const int *arcAlignOff;              /* [nArcs + 1] */
   const int *alState, *alModel, *alDur;
   const float *alLike;
   int models;
} htkamd_lattice_align;
#define HTKAMD_LAT_ALDUR  0x0100
#define HTKAMD_LAT_ALLIKE 0x0200
#define HTKAMD_LAT_DEFAULT_ALIGN 0x03f8
int  htkamd_lattice_write_align(const htkamd_lattice *lat, const htkamd_lattice_align *al, const htkamd_mmf *hmms, const htkamd_net *net, const char *path,
                                const char *utterance, const char *lmName, const char *vocabName, int format);
/* TranscriptionFromLattice's label list for ONE alternative (arcs from htkamd_lattice_nbest) whose arcs carry alignment records
 * (HRec.c:2284-2338): model labels (-m), state labels (-f; with -m too the model rides as their first auxiliary label), the word as the last
 * auxiliary label of an arc's first label.  *out = NULL when an arc of the alternative has no records (word labels are then the caller's). */
int  htkamd_lattice_align_trans(const htkamd_lattice *lat, const htkamd_lattice_align *al, const htkamd_mmf *hmms, const htkamd_net *net,
                                const int *arcs, int nArcs, htkamd_trans **out);
/* TranscriptionFromLattice (HRec.c:2176) with N > 1: the N most likely paths with distinct word sequences, best first.  Alternative i has
 * altLen[i] arcs altArcs[i*maxLen + 0..], start to end, the closing arc into the end node left out. */
int  htkamd_lattice_nbest(const htkamd_lattice *lat, const htkamd_net *net, int N, int maxLen, int *nAlt, int *altLen, int *altArcs);
float htkamd_lattice_arc_score(const htkamd_lattice *lat, int arc);      /* LArcTotLike (HNet.h:257): the score a label of that word carries */
int  htkamd_net_pron_num(const htkamd_net *n, int pron);                  /* v= : number of the pronunciation within its word, from 1 */
void htkamd_decoder_destroy(htkamd_decoder *d);
/* The order in which equally likely tokens reach a node.  SetEntryState (HRec.c:1303) keeps the FIRST of two tokens of exactly equal
 * likelihood, and "first" is a matter of HRec's instance list, which AttachInst / MoveToRecent / ReOrderList / DetachInst
 * (HRec.c:1123-1301) re-order as the utterance goes.  The batch kernel pulls in a static order and NOTES when two such tokens with
 * different histories meet; HTKAMD_ORDER_AUTO (default) then decodes that utterance again with the list itself kept and walked on the
 * device (exact: HVite's choice among homophones of equal score, at a fraction of the batch kernel's speed); _FAST keeps the static
 * order's answer (same likelihoods, possibly the other of two equally scored words); _EXACT sends every utterance through the list
 * kernel.  The environment variable HTKAMD_DECODE_ORDER = auto | fast | exact overrides.  decoder_last_tied: how many utterances of the
 * last htkamd_decoder_run took the list kernel.  N-best runs (htkamd_decoder_run_lattice) always use the list. */
#define HTKAMD_ORDER_AUTO  0
#define HTKAMD_ORDER_FAST  1
#define HTKAMD_ORDER_EXACT 2
int  htkamd_decoder_set_order(htkamd_decoder *d, int mode);
int  htkamd_decoder_last_tied(const htkamd_decoder *d);
/* Device time of the last htkamd_decoder_run: the scoring kernels (K1 + the score block's transposition) and the token kernel(s), milliseconds. */
int  htkamd_decoder_last_times(const htkamd_decoder *d, double *scoreMs, double *tokenMs);
/* Model-instance steps of the last htkamd_decoder_run's token kernel (the register-resident models of the network): out[0] with a live
 * token (StepHMM1 ran: HRec.c:642), out[1] with none (the instance would not be in HRec's list: DetachInst HRec.c:1330).  out[0] 3 states
 * is what demand-driven scoring (HRec.c:438-551) would evaluate against the dense block's tied states x frames. */
int  htkamd_decoder_last_live(const htkamd_decoder *d, long long out[2]);
int  htkamd_decoder_run(htkamd_decoder *d, const htkamd_decode_config *cfg, const float *dX, const int *frameOff, int nUtt,
                        int maxWords, int *nWords, int *wordPron, int *wordStart, int *wordEnd, float *wordScore, float *wordLm,
                        double *total, void *stream);

/* The same with every per-word quantity of the 1-best path that CompleteRecognition's lattice carries (LatFromPaths HRec.c:1512):
   wordAc = LArc.aclike (acoustic log likelihood of the word's segment, float), wordLike = Path.like (the token's likelihood at the
   word end, double).  Any of wordLm / wordAc / wordLike may be NULL. */
typedef struct {
   int *nWords, *wordPron, *wordStart, *wordEnd;
   float *wordScore, *wordLm, *wordAc;
   double *wordLike, *total;
   float *finalLm;             /* [nUtt] LM log probability the final token collected after the last word end (may be NULL) */
} htkamd_decode_out;
int  htkamd_decoder_run_out(htkamd_decoder *d, const htkamd_decode_config *cfg, const float *dX, const int *frameOff, int nUtt,
                            int maxWords, const htkamd_decode_out *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Waveform -> MFCC(+_0/_E)(+_D)(+_A)(+_Z) on the device: replaces what OpenBuffer (HParm.h, HParm.c:4357)
 * does for a waveform source with maxObs == 0 (whole file converted into a table):
 *   GetWave HWave.c:1683 frame slicing; ConvertFrame HParm.c:2214 = PreEmphasise HSigP.c:134, Ham :122,
 *   Wave2FBank :558 (Realft :362 / FFT :311, mel binning, log), FBank2MFCC :607, FBank2C0 :647,
 *   WeightCepstrum :773, raw log energy; NormaliseLogEnergy :911; AddQualifiers HParm.c:1618 =
 *   AddRegression HSigP.c:860 for deltas/accelerations, FZeroMean :803 for _Z.
 * Configuration names follow the HParm config variables (HParm.c:258-367).
 * Output rows have the reference's column order: c1..cN [c0] [E] [deltas] [accs]; the output buffer is a
 * feature matrix that htkamd_outp_block / htkamd_fb_* / htkamd_viterbi_* take as dX.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
   double sampPeriod;          /* SOURCERATE (100 ns units; 625 = 16 kHz)                 */
   double winDur, frPeriod;    /* WINDOWSIZE, TARGETRATE (100 ns units)                    */
   int    numChans, numCeps, cepLifter;                        /* NUMCHANS NUMCEPS CEPLIFTER */
   float  preEmph;                                             /* PREEMCOEF                 */
   int    useHam, usePower, zMeanSource, rawEnergy, eNormalise;/* USEHAMMING USEPOWER ZMEANSOURCE RAWENERGY ENORMALISE */
   float  loFreq, hiFreq, cepScale, silFloor, eScale;          /* LOFREQ HIFREQ (<0 off) CEPSCALE SILFLOOR ESCALE */
   int    hasC0, hasE, hasD, hasA, hasZ;                       /* qualifiers of TARGETKIND   */
   int    delWin, accWin;                                      /* DELTAWINDOW ACCWINDOW      */
} htkamd_mfcc_config;

typedef struct htkamd_mfcc htkamd_mfcc;
int  htkamd_mfcc_create(const htkamd_mfcc_config *cfg, htkamd_mfcc **out);
void htkamd_mfcc_destroy(htkamd_mfcc *f);
int  htkamd_mfcc_num_frames(const htkamd_mfcc_config *cfg, int nSamples);     /* FramesInWave HWave.c:1663 */
int  htkamd_mfcc_num_cols(const htkamd_mfcc_config *cfg);
/* dWav: device int16 samples of nUtt waveforms back to back; sampOff host [nUtt+1]; frameOff host OUT [nUtt+1];
   dOut: device float [frameOff[nUtt] * cols].  Asynchronous on `stream` except for the table upload of the first call. */
/* AddQualifiers (HParm.c:1618 -> AddDiffs :1552 -> Regress HSigP.c:827) on an already parameterised table, e.g. MFCC_E
   files read with TARGETKIND = MFCC_E_D (HTKDemo/toolconfs/herest.conf): dStatic [F x nStat] -> dOut [F x nStat*(1+D+A)],
   regression windows per utterance (frameOff host [nUtt+1]). */
int  htkamd_parm_add_qualifiers(const float *dStatic, const int *frameOff, int nUtt, int nStat, int hasD, int hasA,
                                int delWin, int accWin, float *dOut, void *stream);
/* The whole qualifier step of AddQualifiers for a table: _D _A _T (third differentials = regression of the accelerations over
   THIRDWINDOW, HParm.c:1675-1681), then _Z (FZeroMean over the first nZeroMean columns per utterance, HParm.c:1700-1726: the base
   coefficients, plus C0 when the kind has _0 and not _N), and _N (the row is handed out without its absolute energy / C0 column
   `nullECol`, ExtractObservation HParm.c:2882-2893; needs _D).  dOut: [F x htkamd_parm_quals_cols(q)]. */
typedef struct {
   int nStat;                       /* static columns incl. C0 / energy                        */
   int nZeroMean;                   /* _Z: leading columns to zero-mean (0 = no _Z)            */
   int hasD, hasA, hasT;            /* _D _A _T                                                */
   int delWin, accWin, thirdWin;    /* DELTAWINDOW ACCWINDOW THIRDWINDOW                       */
   int nullECol;                    /* _N: column left out of every row (-1 = none)            */
   int v1Compat, simpleDiffs;       /* V1COMPAT (edge rows = plain differences), SIMPLEDIFFS    */
} htkamd_parm_quals;
int  htkamd_parm_quals_cols(const htkamd_parm_quals *q);
int  htkamd_parm_qualify(const float *dStatic, const int *frameOff, int nUtt, const htkamd_parm_quals *q, float *dOut, void *stream);
/* The same qualifiers in HParm's BUFFER mode (OpenBuffer / ReadAsBuffer, HParm.c:4000-4116 FillBufFromChannel; HVite.c:664 opens its
   input this way): static rows arrive in pushes, an observation is handed out as soon as the qwin = DELTAWINDOW + ACCWINDOW
   (+ THIRDWINDOW) rows of look-ahead its regression windows need have arrived; the last push (last = 1) flushes the remaining rows
   with the end-of-utterance replication.  The rows handed out are bit-identical to htkamd_parm_qualify over the whole utterance.
   _Z is refused (it needs the whole utterance: table mode), as HParm refuses ENORMALISE on a live buffer (HParm.c:4096).
     open : maxRows = the largest push;   push : dStatic [nRows x nStat] on the device, dOut receives *nOut <= nRows + qwin rows. */
typedef struct htkamd_parm_stream htkamd_parm_stream;
int  htkamd_parm_stream_open(const htkamd_parm_quals *q, int maxRows, htkamd_parm_stream **out);
int  htkamd_parm_stream_push(htkamd_parm_stream *s, const float *dStatic, int nRows, int last, float *dOut, int *nOut, void *stream);
int  htkamd_parm_stream_lookahead(const htkamd_parm_stream *s);
void htkamd_parm_stream_close(htkamd_parm_stream *s);
/* HCompV's global statistics over a device table of frames: replaces AccVar / CalcCovs (HTKTools/HCompV.c:392-411, :261-291) --
   mean and diagonal variance of all frames, variance floored at minVar (HCompV -v, default 0).  The reference sums in float in
   file order; this sums in fp64 (order-free), so the results agree to the reference's own rounding (~1e-6 relative at a few
   thousand frames, growing with the frame count on the reference's side).  Flat start = every Gaussian's mean/variance set to
   these (HCompV -m), the variance floor macro = f * var (HCompV -f f, htkamd_mmf_write_vfloors). */
int  htkamd_compv(const float *dX, long long nFrames, int D, float minVar, float *mean /*[D]*/, float *var /*[D]*/, void *stream);
int  htkamd_mfcc_compute(htkamd_mfcc *f, const short *dWav, const int *sampOff, int nUtt, int *frameOff, float *dOut, void *stream);

#ifdef __cplusplus
}
#endif
#endif
