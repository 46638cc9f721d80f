// decode.h -- what the network decoder's kernels (decode.hip: 1-best; decode_n.hip: N-best token sets + lattice) and their host code share.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "internal.h"

#define DEC_THREADS 1024
#define DEC_MAXN 8              /* states per model incl. entry/exit */
#define DEC_WIDE 96             /* fan-in from which a node is reduced by the whole workgroup */

struct DecNet {
   int nNodes, nHmm, nLevels, nWordNodes, initial, final, nTok, nTpFloats;
   const int *kind, *model;            // [nNodes]
   const float *pronProb;              // [nNodes]
   const int *predOff, *predSrc;       // reverse CSR; bit 31 of predSrc: the predecessor is a word/null node
   const float *predLike;
   const int4 *nodeInfo;               // [nNodes] {kind | N << 4 | tee << 12, first token, offset of transP, offset into hmmState}
   const int *tok0;                    // [nNodes] first token (state 1) of the node
   const int *hmmNodes;                // [nHmm] model nodes (tee or not)
   const int *nodeN, *nodeTp, *nodeSt; // [nNodes] HMM: numStates, offset of transP, offset into hmmState
   const unsigned char *nodeTee;       // [nNodes]
   const float *wdlk;                  // [nNodes]
   const int *wordIdx;                 // [nNodes] dense index of WORD nodes (path table column) or -1
   const int *wordNode;                // [nWordNodes] inverse of wordIdx
   const int *levelOff, *levelNodes;   // zero-time nodes by level: narrow ones first, then wide ones
   const int *levelWide;               // [nLevels] index in levelNodes where the wide nodes of the level start
   const int *levelOffAll, *levelNodesAll, *levelWideAll;   // the same with the nodes k_decode<NPT > 0> steps from registers (regFused) left in
   const float *transP;
   const int *hmmState;
   const int *stateSlot;               // [S] row of the tied state in the score block, -1 if unused
   // forward CSR in the network's own link order (the tr0 links first, ExpandWordNet HNet.c:3632-3645): the exact-order kernels
   // (decode_ord.hip) PUSH along it as HRec does
   const int *linkOff, *linkDest;      // [nNodes + 1], [nLinks]
   const float *linkLike;
   const int *nTr0;                    // [nNodes] number of leading links whose destination is a zero-time node (word end, null node, tee model)
   const unsigned char *dupDest;       // [nNodes] two links of the node share a destination
   // k_decode<NPT > 0>: hmmNodes[0 .. nReg) are plain models of at most three emitting states whose tokens live in registers
   int nReg;
   const int *regFused;                // [nReg] the word / null node whose only predecessor is this model (stepped by its owner), or -1
   const float *regFusedLike;          // [nReg] LM log probability of that link
   const unsigned char *regNoEx;       // [nReg] nobody pulls this model's exit token from memory
   // k_decode<.., EXL>: the exit tokens the register-resident models pull are kept in LDS beside memory -- a WORD node's as its likelihood alone
   // (its lm is 0 and its path is frame * nWordNodes + wordIdx: StepWord2 HRec.c:1046), a null node's whole
   const int *zl;                      // [nNodes] -1, or bit 30 | wordIdx (WORD node), or the null node's slot in the LDS table
   int nNullLds;                       // null nodes with a slot (<= DEC_NULL_LDS)
   const int2 *predRecL;               // [links] predRec with x = bit 31 | bit 30 | wordIdx for a WORD predecessor, bit 31 | bit 29 | slot for a null node with a slot
   const int *regFusedZl;              // [nReg] zl of the model's fused node (-1: none / no slot)
   const int4 *regFusedRec;            // [nReg] {fused node or -1, its zl, bits of its pronProb, its wordIdx or -1}: StepWord2 without four dependent loads
   const int2 *regPred;                // [nReg] the model's range in predRecReg / predRecRegL: the entry pull's range without the two loads behind the node's number
   const int2 *predRecReg, *predRecRegL;   // the register-resident models' lists of predRec / predRecL, in the models' order
   const int2 *predRec;                // [links] {predSrc, bits of predLike}: one load per predecessor
   const int4 *regRecA, *regRecB; const float2 *regRecF;      // [nReg] the same and the model's constants packed: {node, kind | N << 4, transP offset, fused node}, {score slots of states 2..4, regNoEx}, {wdlk, fused link's LM}
};
// 1024 threads x 6 models: four wavefronts per SIMD hide one another's memory latencies; 512 x 12 (round 4: no register spills to speak of,
// two wavefronts per SIMD) took 50 ms where this takes 41 on the 6 000-word bigram network, 768 x 8 42 (tools/r05_decvar.sh)
#define DEC_REG_THREADS 1024
#define DEC_REG_MAXNPT 6
#define DEC_NULL_LDS 64             /* null nodes whose tokens k_decode<.., EXL> keeps in LDS */

struct DecUtt {
   int T, frame0, status, idx;         // idx: the utterance's number in the batch (its slot in the per-utterance outputs)
   size_t score0;      // floats: score[score0 + slot*T + (t-1)]
   size_t tok0;        // token arrays base
   size_t node0;       // exit-token / instance-max arrays base
   size_t path0;       // path table base: (t*nWordNodes + w)
   size_t out0;        // word output base
};

struct __attribute__((aligned(16))) Tok { double like; float lm; int path; };   // one 16-byte load/store per token

struct DecArgs {
   DecNet net;
   const DecUtt *utt; int nUtt;
   const float *score; int ns;         // frame-major since round 4: score[score0 + (t-1)*ns + slot] (k_score_transpose), ns = tied states of the network
   Tok *tok;                           // [sum nTok]   state tokens
   Tok *ex; double *imax;              // [sum nNodes] exit tokens, instance maxima
   int *pathPrev; double *pathLike; float *pathLm;
   float genBeam, wordBeam, lmScale, wordPen, prScale;
   int maxActive;                      // HVite -u: maximum number of model instances kept per frame (0 = off)
   int maxWords;
   int *nWords, *wordPron, *wordStart, *wordEnd; float *wordScore, *wordLm, *wordAc; double *wordLike; double *total; float *finalLm;
   unsigned long long *liveCnt;        // [nUtt][2] k_decode<NPT > 0>: register-resident model instances stepped with a live token / with none (htkamd_decoder_last_live)
   int *tieFlag;                       // [nUtt] k_decode: two tokens of exactly equal likelihood and different histories met at a node (see decode_ord.hip)
};

// Exact-order decoding (decode_ord.hip): the utterances k_decode flagged, or every utterance of an N-best run.
struct OrdArgs {
   DecArgs d;                          // utt = the selected utterances' descriptors (idx = slot in the outputs)
   int *seq; int seqCap;               // [nSel * 2 * seqCap] the instance list as an array (two buffers per utterance)
   int *pos; unsigned char *ooo;       // [sum nNodes] (at DecUtt.node0) position of the node's instance in seq (-1: none); NetInst.ooo
   int *pathNode, *pathFrame;          // [paths] (at DecUtt.path0) Path records are allocated one by one here
   int pathExtra;                      // records per utterance beyond (T + 1) * nWordNodes
   int keepFast;                       // the outputs hold k_decode's result of the same utterance (HTKAMD_ORDER_AUTO): a walk that runs out of one of its
                                       // fixed capacities (-4 path records, -5 list appends of a frame, -6 nesting of zero-time nodes) leaves it standing
};
int htkamd_launch_decode_ord(const OrdArgs &a, int nSel, hipStream_t s);
// K1 writes a score block state-major (a state's frames are a lane's stores); the token loops read a COLUMN per frame: one 128-byte line per
// state and frame -- 640 KB per utterance and frame on the 5k-state set.  Transposed once (tiles through LDS), a frame's column is 20 KB.
int htkamd_launch_score_transpose(const float *in, float *out, const DecUtt *dUtt, int nUtt, int maxT, int ns, hipStream_t s);


struct htkamd_decoder {
   htkamd_model *m;
   DecNet net;                         // device pointers
   std::vector<void *> owned;
   std::vector<int> usedStates;        // tied states of the network in slot order
   std::vector<int> hostModel;         // [nNodes] htkamd_net_desc.model (WORD nodes: the pronunciation)
   int *d_usedStates;
   int maxWidthNodes;
   // workspace of htkamd_decoder_run, kept between calls and grown when a batch needs more (the score block and the path tables are
   // gigabytes at 256 utterances: allocating and freeing them per call cost 5 - 500 ms of a 130 ms call)
   void *ws[32];
   size_t wsCap[32];
   int orderMode;                      // HTKAMD_ORDER_AUTO / _FAST / _EXACT (htkamd_decoder_set_order)
   int lastTied;                       // utterances of the last run that went through the exact-order kernel
   hipEvent_t ev[4];                   // around the scoring kernels and around the token kernel of the last chunk of a run
   float lastScoreMs, lastTokenMs;
   long long lastLive[2];              // model-instance steps of the last run's register kernel: with a live token, without
   void *wsN[48];                      // ... and of htkamd_decoder_run_lattice
   size_t wsNCap[48];
};
