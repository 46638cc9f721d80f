/* internal.h -- shared declarations of the htk_amd native library (not part of the C ABI). */
#ifndef HTKAMD_INTERNAL_H
#define HTKAMD_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include "../../include/htk_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HTK constants (HMath.h:42-45, HModel.h:52-53) */
#define LZERO    (-1.0E10)
#define LSMALL   (-0.5E10)
#define MINEARG  (-708.3)
#define MINLARG  2.45E-308
#define HTK_TPI  6.28318530717959
#define MINMIX   1.0E-5
#define LMINMIX  (-11.5129254649702)

void htkamd_set_error(const char *fmt, ...);

/* ---- host-side preparation (htk_amd/host/prep.c) ---- */
void   htkamd_host_fix_diag_gconst(int D, const float *var, float *gconst);   /* HModel.c:5641 */
void   htkamd_host_conv_diagc(size_t n, const float *var, float *ivar);        /* HUtil.c:413  */
float  htkamd_host_mix_log_weight(float w);                                    /* HModel.c:5288 */
int    htkamd_host_min_dur(int N, const float *tp);                            /* HFB.c:106    */
int    htkamd_host_stream_dims(const char *kind, int vecSize, int S, const int *width, int *dimStream, char *why, size_t whyLen);   /* HParm.c:3094,2843 */
void   htkamd_host_fix_diag_gconst_ms(int D, const float *var, const int *dimStream, int stream, float *gconst);
int    htkamd_host_trans_is_lr(int N, const float *tp);                        /* left-to-right, no skips (fb_lr.hip) */
double htkamd_host_min_log_exp(void);                                          /* HMath.c:1680 */

/* device LAdd table: 4 intervals per unit of d over [minLogExp, 0] = [-23.03, 0], degree-10 Taylor rows (8 KB) */
#define LADD_INV_H 4
#define LADD_DEG   10
#define LADD_NK    93           /* floor(23.0259 * 4) + 1 */
int    htkamd_host_ladd_table_size(void);
void   htkamd_host_build_ladd_table(double *tab);

/* ---- MFCC front-end tables (htk_amd/host/fbank.c) ---- */
struct htkamd_mfcc_tables {
   int frSize, frRate, fftN, klo, khi;
   float mfnorm;
   float *ham;                 /* [frSize+1] 1-based */
   float *cepWin;              /* [numCeps+1] */
   float *loWt;                /* [fftN/2+2] 1-based */
   int *binA0, *binA1, *binB0, *binB1;   /* [numChans+2] k ranges feeding each bin (one allocation at binA0) */
   double *dct;                /* [(numCeps+1)*(numChans+1)] cos(x_j*(k-0.5)) */
   double *tw;                 /* FFT twiddles (wr,wi), stages concatenated: fftN/2 pairs */
   double *rtw;                /* Realft (yr,yi) for i = 2..fftN/4 at [2i],[2i+1] */
   short *brev;                /* [fftN/2] bit-reversed complex index */
};
int  htkamd_mfcc_tables_build(const htkamd_mfcc_config *c, struct htkamd_mfcc_tables *t);
void htkamd_mfcc_tables_free(struct htkamd_mfcc_tables *t);

/* ---- packed model ---- */
struct htkamd_model {
   int D, S, C, G, nT, H, maxN, maxM;
   int PS;                     /* floats per Gaussian in gparam: 2*D+1 rounded up to a multiple of 4 */
   /* host copies */
   int   *h_stateCompOff, *h_compGauss, *h_transN, *h_transOff, *h_hmmTrans, *h_hmmStateOff, *h_hmmState, *h_minDur;
   int   *h_trOccOff;          /* [nT+1] prefix sum of transN */
   unsigned char *h_transLR;   /* [nT] the matrix is left-to-right without skips: a_1j only for j = 2, a_ij only for j = i, i+1, a_iN only from N-1
                                  (kept with h_minDur; a change bumps topoVersion) */
   float *h_mean, *h_var, *h_ivar, *h_gconst, *h_compWeight, *h_compLogWt, *h_transP;
   /* device copies */
   float *d_gparam;            /* [G*PS]: (mean[i], ivar[i]) pairs, 8-byte aligned, then gconst at [2*D] */
   double *d_laddTab;          /* [LADD_NK*(LADD_DEG+1)] */
   float *d_mean, *d_ivar, *d_gconst, *d_compLogWt, *d_transP;
   int   *d_stateCompOff, *d_compGauss, *d_transN, *d_transOff;
   /* device-side update (update.hip): linear parameters and topology, uploaded on first use */
   float *d_var, *d_compWeight;
   int   *d_trOccOff, *d_hmmTrans, *d_hmmStateOff, *d_hmmState;
   void  *d_updScratch; size_t updScratchCap;
   void  *evUpd; int updPending;  /* htkamd_model_update_device_begin .. _end: the event behind the update's copy */
   void  *h_updPin;            /* pinned: transition matrices + the 16 counters of a device update, one D2H copy */
   int    topoVersion;         /* bumped whenever a minimum duration (hence tee-ness) of a transition matrix changes: batch tables
                                  prepared before that (htkamd_fb_prepare) no longer describe the model */
   void  *obRing;              /* task-table ring of htkamd_outp_block (gmm_exact.hip) */
   int    hostStale;           /* the host copies of mean/var/gconst/weights are older than the device's (after htkamd_model_update_device) */
   /* MFMA scoring path (gmm_mfma.hip): A-operand fragments [tile][mfmaNS+4][64], 16 components per tile */
   float *d_mfmaTab;           /* NULL when D has no MFMA kernel */
   int   *d_stateTileOff;      /* [S+1] */
   int   *d_tileState;         /* [nTiles] tied state of every fragment tile */
   int    mfmaNS, nTiles;
   int    mfmaStale;           /* the fp32 fragment table is older than the parameters (device update): rebuilt on its next use */
   void  *d_bf16Tab;           /* bf16 x 3 scoring path (gmm_bf16.hip): A-operand pieces per tile; NULL when D > 45 */
   int    bf16NC;              /* K chunks of 32 per piece: ceil(D/15) */
   int    bf16Dense;           /* the 32 x 32 bf16 kernel's five-k-step layout (31 <= D <= 39, every state in one tile): gmm_bf16.hip, k_score_bf16w<5> */
   void  *d_f16Tab;            /* fp16 x 2 scoring path (gmm_f16.hip): A-operand pieces per tile, K chunks as the bf16 path; NULL when D > 45 */
   int    f16Wide;             /* every state fits one tile (<= 16 components): the 32 x 32 form of the kernel and its table layout */
   float *d_f16Ctl;            /* its control block: scale[96], 1/scale[96], range[192], flag of the last table build, sticky range flag */
   int    bf16Stale, f16Stale; /* the table is older than the parameters (a device update while the path was not in use): rebuilt on its next use */
   int    compat;              /* HTKAMD_COMPAT_* bits (htkamd_model_set_compat) */
   unsigned char *h_rawLogWt, *d_rawLogWt;   /* [C] HTKAMD_COMPAT_SHARED_LOGWT: the component's weight is read as a LOG weight as it stands (ConvLogWt skipped it), or NULL */
   int    fastUse;             /* HTKAMD_SCORE_BF16 / _F16 bits: the paths that have scored with this model (their tables follow every device update) */
   /* shared mean / variance vectors (~u / ~v macros; htkamd_model_set_sharing): first Gaussian of the group a Gaussian's mean / variance
      belongs to (itself when private), members of its variance group; NULL = no sharing in the set */
   int   *h_meanLeader, *h_varLeader, *h_varGroupSize;
   int   *d_shareTab;                 /* meanLeader[G] varLeader[G] varGroupSize[G] muMemOff[G+1] vaMemOff[G+1] muMem[] vaMem[] (update.hip) */
   int    shareMuMem, shareVaMem;     /* lengths of the two member lists */
   /* several streams (htkamd_model_desc::numStreams > 1): S counts (state, stream) ELEMENTS, element = tied state * NSt + stream, and
      h_hmmState lists the NSt elements of every emitting state; a Gaussian is an undivided row whose dimensions outside its stream
      carry mean 0, 1/variance 0 */
   int    NSt;                 /* streams (1: none of the following is allocated) */
   int   *h_dimStream, *h_gaussStream, *d_dimStream, *d_gaussStream;   /* [D] stream of a dimension, [G] stream of a Gaussian */
   /* tied mixtures (htkamd_model_desc::hsKind == HTKAMD_HS_TIED): the pool of stream k is what element k (state 0) lists */
   int    tiedMix, tmPool;     /* tmPool: Gaussians in all pools together */
   float  tmBeam;              /* aligner / decoders: PrecomputeTMix's threshold (HVite -c, default 10.0; htkamd_model_set_tm_beam) */
   int   *h_tmPoolOff, *d_tmPoolOff;   /* [NSt+1] first pool entry of a stream in the per-frame pool table (fb.hip tmE) */
   float *h_streamWt, *d_streamWt;    /* [S] stream weight of every element (1 unless <SWEIGHTS>) */
   int   *d_msCompOff;         /* [S+1] = 2e: what the recursion kernels read as "components of the chain state" (they only ask == 1) */
   int   *h_scanOrder;         /* [H] the reference's HMM scan order of the physical models (htkamd_model_set_scan_order), or NULL */
   double minLogExp;
};

void htkamd_outp_ring_free(void *ring);                      /* gmm_exact.hip */
int htkamd_model_refresh_mfma_device(struct htkamd_model *m, void *stream);   /* update.hip */
int htkamd_model_refresh_bf16_device(struct htkamd_model *m, void *stream);   /* gmm_bf16.hip (stream: hipStream_t) */
int htkamd_model_refresh_f16_device(struct htkamd_model *m, void *stream);    /* gmm_f16.hip */
int htkamd_model_f16_flag(struct htkamd_model *m, void *stream, int *flag);   /* gmm_f16.hip: the sticky range flag, read and cleared */
/* range flag of the fp16 path (ScoreArgs::rangeFlag) */
#define HTKAMD_F16_EMODEL 1    /* a scaled coefficient of the model exceeds fp16's range */
#define HTKAMD_F16_EFEAT  2    /* a scaled feature value (x or x^2) exceeds fp16's range */
int htkamd_model_device_tables(struct htkamd_model *m);      /* model.hip: uploads d_var etc. once */
int htkamd_model_sync_host(struct htkamd_model *m);          /* model.hip: device -> host parameter copies when stale */
int htkamd_update_models(struct htkamd_model *m, const htkamd_accs_layout *lay, const double *acc,
                         const htkamd_update_config *cfg, htkamd_update_stats *st);   /* host/update.c */

struct htkamd_accs {
   struct htkamd_model *m;
   htkamd_accs_layout lay;
   double *d_vec;
   int stateOrder;             /* htkamd_accs_state_ranges: 0 not looked at yet, 1 compGauss[c] == c throughout (a state's Gaussians are a range of the vector), -1 not so */
};

#ifdef __cplusplus
}
#endif
#endif
