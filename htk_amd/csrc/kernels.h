// kernels.h -- device-side argument blocks shared by the launchers (internal).
#ifndef HTKAMD_KERNELS_H
#define HTKAMD_KERNELS_H
#include <hip/hip_runtime.h>
#include "internal.h"

#define SCORE_TILE_FRAMES 128   /* 64 lanes x 2 frames per lane */
#ifndef B16_TASK_FRAMES
#define B16_TASK_FRAMES 128     /* frames per task of the bf16 matrix-core kernel in forward-backward (experiment switch HTKAMD_B16_TF) */
#endif
#define SCORE_TASK_SLOTS  16    /* chain states scored per task */
#define SCORE_TASK_SLOTS_WIDE 64 /* the same for the matrix-core kernels in forward-backward (fb.hip) */
#define SCORE_TASK_SLOTS_EXACT 8 /* and for the exact kernel there: one wave per task, features in registers, so small tasks balance best
                                    (5.28 / 5.19 / 5.11 ms at 16 / 12 / 8 on the bench workload) */

struct ScoreTask {
   int frame0;        // first row of the tile in X
   int nFrames;       // valid frames in the tile (<= SCORE_TILE_FRAMES)
   int slot0;         // first entry in slotState
   int nSlots;        // states to score
   int outSlot0;      // row of the first state in the output block
   int ldo;           // leading dimension (frames) of the output block
   size_t outBase;    // element offset of (row 0, first frame of this tile) in out
};

struct ScoreArgs {
   const ScoreTask *tasks;
   int nTasks;
   const float *X;
   const int *slotState;
   float *out;
   const int *stateCompOff, *compGauss;
   const float *compLogWt, *gparam;
   int PS, D;
   double minLogExp;
   const double *laddTab;
   int *taskCounter;          // dynamic task queue head (zeroed before the launch)
   const float *mfmaTab;      // MFMA path only
   const int *stateTileOff;
   const void *bf16Tab;       // bf16 x 3 path only
   const void *f16Tab = nullptr;       // fp16 x 2 path only (set by its launcher): the table, the model's control block
   const int *f16Ctl = nullptr;
   int *rangeFlag = nullptr;           // fp16 x 2 path: where HTKAMD_F16_* bits are raised (NULL: the model's sticky flag)
   const float *var;          // DIAGC form only: variances [G*D]
   // several streams, HRec's state output probability (cPOutP HRec.c:510-548: outp += w[s] * cSOutP(s), float): a slot names the state's
   // first (state, stream) element, the kernel scores its NSt elements and writes their weighted sum.  NSt <= 1 (what forward-backward
   // passes, whose rows ARE elements): a slot is scored as it stands.
   int NSt = 1;
   const float *streamWt = nullptr;   // [elements]
   // k_score_bf16w: the task list cut into eight queues, one per XCD (tasks of utterance u in queue u % 8, so that the frame tiles of a
   // state chunk -- which stream the same 64 table tiles -- are pulled by workgroups behind the same L2); qStart[9], qCounters[8]; NULL: one queue
   const int *qStart = nullptr;
   int *qCounters = nullptr;
   // forward-backward, k_score_bf16w / k_score_f16w: per entry of slotState the first and the last row of X in which Setotprob of the un-pruned
   // pass evaluates the chain state (HFB.c:1014, 1177, 1215) -- [2] ints per slot; NULL (block scoring, decoders): every frame of a task
   const int *slotRange = nullptr;
};

// evStart/evStop (may be NULL): updated with the dispatch's own start and stop time (hipExtLaunchKernel), i.e. without the time
// the kernel waits for the machine when another stream is using it
int htkamd_launch_score_exact(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr, bool soutp = false, bool diagc = false);
int htkamd_launch_score_mfma(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr);
int htkamd_launch_score_bf16(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr);
int htkamd_launch_score_f16(const htkamd_model *m, const ScoreArgs &a, hipStream_t stream, hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr);
// the scoring kernel of a score mode (HTKAMD_SCORE_* bits: F16 before BF16 before MFMA before exact)
static inline int htkamd_launch_score(int mode, const htkamd_model *m, const ScoreArgs &a, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
   if (mode & HTKAMD_SCORE_F16) return htkamd_launch_score_f16(m, a, s, e0, e1);
   if (mode & HTKAMD_SCORE_BF16) return htkamd_launch_score_bf16(m, a, s, e0, e1);
   if (mode & HTKAMD_SCORE_MFMA) return htkamd_launch_score_mfma(m, a, s, e0, e1);
   return htkamd_launch_score_exact(m, a, s, e0, e1, (mode & HTKAMD_SCORE_SOUTP) != 0, (mode & HTKAMD_SCORE_DIAGC) != 0);
}

// ---- forward-backward ----
struct UttDesc {
   int T, Q, nCells, nSlots;
   int frame0;        // first row in X and in the per-frame beam arrays
   int q0;            // first entry in the per-model tables
   int cell0;         // first entry in the per-cell tables
   int slot0;         // first entry in slotState
   int status;        // host pre-check (CreateInsts): HTKAMD_UTT_*
   int nEval;         // output-probability evaluations of the un-pruned pass (metric unit)
   int thr0, nThr;    // this utterance's slice of thrCell; threads in use
   size_t outp0;      // floats : outp[outp0 + slot*T + (t-1)]
   size_t beta0;      // doubles: beta[beta0 + (t-1)*nCells + cell]
   size_t gam0;       // doubles: gam [gam0  + (t-1)*nSlots + slot]
   size_t betaW0;     // doubles: wave path's beta block of this utterance, betaW[betaW0 + ((t-1)*5 + i-1)*64*W + model-1]
   int W, pad;        // wavefronts working on the utterance (1, 2, 4 or 8); 0 = general kernels.  pad = 0: a lane per MODEL (fb_wave.hip,
                      // chain of <= 64*W models); pad = 1: a lane per chain STATE (fb_state.hip, <= 64*W emitting states, no tee models),
                      // beta block = betaS[T][64*W] then betaE[T][64*W]; pad = 2: the same for left-to-right chains (fb_lr.hip), beta block =
                      // betaS[T][64*W], alpha block (alphaW0) = alphaS[T][64*W] then alphaE[T][QP]
   size_t alphaW0;    // doubles: the utterance's block in alphaW (pad = 2)
   int QP, pad2;      // Q rounded up to a multiple of 8
};

// what the statistics of a (frame, chain state) pair need of the state (left-to-right path: written by the beta kernels, a record per slot)
struct LaneRec { float aSelf, aOut, aEntry, aEntryNext; short q, j, N, pad; int sidx, cM; };
struct HitS { int frame, pad; double seed; };      // a surviving (frame, state) pair in the list bucketed by tied state (k_mixstate): the state is the bucket
struct MixRec { int g, frame; double L; };          // posterior L of Gaussian g at row `frame` of the feature table
struct MixHit { int st, frame; double seed; };      // a (frame, state) pair the MINFORPROB prune lets through: tied state, row of the feature table, seed

struct FbArgs {
   const UttDesc *utt;
   int nUtt;
   const int *uttList;               // the utterances this launch works on (one class: same W), and their number
   int nList;
   // per-model tables (index q0 + q - 1)
   const int *mN, *mTp, *mCell0, *mSlot0, *mDms, *mHmm, *mTrans;
   // per-cell tables (index cell0 + c)
   const short *cQ, *cI;
   const short *thrCell;             // [thr0 + thread] -> cell (or -1): threads grouped by role, padded to waves
   const int *slotState;
   const short *sQ;                  // [slot0 + slot] -> model (1-based) of the chain state (state-per-lane kernels)
   // per-frame (index frame0 + t - 1)
   const short *taperLo, *taperHi;
   short *qLo, *qHi, *aLo, *aHi;
   const float *X;
   const float *transP;
   float *outp;
   double *beta, *gam, *alphaDbg;    // alphaDbg: NULL unless debugging, layout as beta
   double *betaW;                    // wave path: beta per utterance as [t-1][state 0..4][64*W lanes] from UttDesc::betaW0 (coalesced per state)
   double *alphaW;                   // left-to-right path (fb_lr.hip): the alpha columns, UttDesc::alphaW0
   int *qBeam, *aBeam;               // left-to-right path: per frame lo | hi << 16 of the beta / alpha beam
   double *trPart;                   // left-to-right path: partial transition counts, a row per (utterance, chunk of frames, wavefront)
   double *pr;                       // [nUtt]
   int *status;                      // [nUtt]
   // model tables for the statistics kernel
   const int *stateCompOff, *compGauss, *transOff, *trOccOff;
   const float *compLogWt, *gparam, *mean;
   const double *laddTab;
   int PS, D, maxN, maxM;
   int nCellsMax, QMax, TMax;        // maxima over the batch (LDS carve)
   double *acc;                      // accumulator vector
   htkamd_accs_layout lay;
   double pruneInit, pruneInc, pruneLim, minLogExp;
   float minFrwdP;
   int uFlags;
   size_t gamTotal;                  // doubles in gam for this batch
   const size_t *gamOffByUtt;        // [nUtt+1] = utt[u].gam0 (for the flat-index -> utterance search)
   const int *gamChunkUtt;           // [ceil(gamTotal/512)] utterance holding seed 512*c
   // statistics kernel, record path: the surviving (frame, Gaussian, posterior) triples are listed, bucketed by Gaussian and summed
   // per Gaussian (one atomic per accumulator element and batch instead of one per triple); NULL = direct atomics only
   MixRec *rec, *recSorted;
   int recCap, G;
   int *recCtl;                      // [0] number of records asked for (may exceed recCap), then gCnt[G+1], gStart[G+1], gCur[G+1]
   // left-to-right path: the surviving (frame, state) pairs as a list (k_stats_lr -> k_mixhits) instead of the dense seed array
   // several streams (model NSt > 1): scores per (stream, chain state) and the map dimension -> stream
   int NSt;
   const float *outpU;               // stream k of utterance u: outpU[NSt*outp0 + (k*nSlots + slot)*T + t-1]
   // HTKAMD_COMPAT_STREAM_REVISIT (k_beta): the next chain state of the same tied state in Setotprob's visiting order seen backwards --
   // same model and an earlier state, else the nearest model further right -- as a slot of the utterance, -1 = none
   int compatRevisit;
   const int *nextSame;              // [slot0 + slot]
   const int *dimStream;
   // tied mixtures (model tiedMix): per frame of the batch the pool's scaled probabilities (-1 = pruned by PrecomputeTMix) and their
   // maximum per stream; the pool-to-state kernel's task list and rows
   float *tmE;                       // [totalFrames][tmPool]
   float *tmMaxP;                    // [totalFrames][NSt]
   const int *tmPoolOff;             // [NSt + 1]
   int tmPool, totalFrames;
   const ScoreTask *tmTasks; int tmNTasks; const int *tmSlotState; float *tmOut;
   const float *compWeight, *var;    // linear weights, variances
   int tmCombine;                    // HVite side: a row is a STATE (its first element), scored as sum_s w_s SOutP_s (cPOutP HRec.c:540); 0: a row is an element
   const float *streamWt;            // [elements], with tmCombine
   MixHit *hits;                     // region r (one per wavefront of k_stats_lr, numbered like the rows of trPart): hits[r * hitRegionCap ...]
   int hitSlots;                     // multi-stream / tied-mixture sets: MixHit::st is the pair's global SLOT (slot0 of its utterance + chain state), not its tied state
   int *hitCtl;                      // [r] records in region r
   int nHitRegions, hitRegionCap;
   LaneRec *laneRec;                 // left-to-right path: [slot0 + lane], written by the beta kernels
   const int *qBeamNP;               // left-to-right path: the beta beams of the un-pruned pass (host: SetBeamTaper alone decides them), lo | hi << 16 per frame
   // mixture statistics bucketed by tied state (k_mixstate; sets of one stream with <= 16 components per state): k_stats_sp drops a surviving
   // pair into its state's bucket (stCnt[st] counts, stBucket[st * stCap ...]); what a bucket has no room for goes to the list of k_mixhits
   int stBase;                                                   // k_mixstate: first tied state of the launch (htkamd_fb_execute_mix takes the states in ranges)
   int *stCnt; int nTiedStates; HitS *stBucket; int stCap;      // stCnt[nTiedStates] = pairs turned away by a full bucket (0: the list kernels have nothing to do)
   int fastMath;                     // the pass runs in the fp32-transcendental class (HTKAMD_SCORE_FASTLADD): posteriors by v_exp_f32
   double *sink;                     // 64 bytes nobody reads: where the lanes outside a beam "store" (one cache line instead of a branch around the store)
   int lrExp;                        // ablation bits of an -DLR_EXP_BUILD=1 build (tools/lr_exp.py); 0 otherwise
};

int htkamd_launch_beta(const FbArgs &a, int blockDim, size_t lds, hipStream_t s);
int htkamd_launch_alpha(const FbArgs &a, int blockDim, size_t lds, hipStream_t s);
// deferState: where the state-bucketed kernel applies, only what its buckets turned away is taken now (by the list kernel, direct atomics);
// the states themselves wait for htkamd_launch_mixstate_range (htkamd_fb_execute_mix)
int htkamd_launch_mixstats(const FbArgs &a, hipStream_t s, bool dense, bool listed, bool deferState = false);
bool htkamd_mixstate_applies(const FbArgs &a);
int htkamd_launch_mixstate_range(const FbArgs &a, int st0, int st1, hipStream_t s);
// several streams: outp = sum over streams of outpU; UpMixParms per stream from the dense seed array
int htkamd_launch_combine_streams(const FbArgs &a, hipStream_t s);
int htkamd_launch_mixstats_ms(const FbArgs &a, hipStream_t s);
// tied mixtures: PrecomputeTMix for every frame, SOutP for every (row, frame) of the task list, UpMixParms' TIEDHS branch
int htkamd_launch_tm_score(const FbArgs &a, hipStream_t s);
int htkamd_launch_mixstats_tm(const FbArgs &a, hipStream_t s);
int htkamd_launch_mixhits_streams(const FbArgs &a, bool tied, hipStream_t s);      // the same two from the lists of the left-to-right path
// aligner / decoders on a tied-mixture set: the score block of `sa` (rows = states) filled by PrecomputeTMix(tmBeam) + SOutP over the first nRows feature rows
int htkamd_tm_score_block(const htkamd_model *m, const ScoreArgs &sa, int nRows, float tmBeam, hipStream_t s);
// wave-per-utterance fast path (fb_wave.hip): chains of <= 64 models with <= 5 states each
// state-per-lane fast path (fb_state.hip): chains of <= 512 emitting states, models of <= 5 states, no tee models
int htkamd_launch_beta_s(const FbArgs &a, int W, bool fast, hipStream_t s);
int htkamd_launch_alpha_s(const FbArgs &a, int W, bool fast, hipStream_t s);
int htkamd_launch_beta_w(const FbArgs &a, int W, bool fast, hipStream_t s);
// left-to-right chains (fb_lr.hip): state-per-lane recursions without statistics + frame-parallel statistics
int htkamd_launch_beta_lr(const FbArgs &a, int W, bool fast, hipStream_t s);
bool htkamd_beta_lr_is_lean(const FbArgs &a, bool fast);      // the pass's beta kernel reads the host's un-pruned beams (FbArgs::qBeamNP) instead of writing FbArgs::qBeam
int htkamd_launch_alpha_lr(const FbArgs &a, int W, bool fast, hipStream_t s);
int htkamd_launch_stats_lr(const FbArgs &a, int W, bool fast, hipStream_t s);
bool htkamd_stats_lr_is_sparse(const FbArgs &a);             // k_stats_sp takes the pass's statistics (it counts the surviving pairs per tied state)
int htkamd_stats_lr_chunks(int TMax);
size_t htkamd_stats_lr_row_doubles(void);
int htkamd_stats_lr_region_cap(void);
int htkamd_launch_alpha_w(const FbArgs &a, int W, bool fast, hipStream_t s);

#endif
