// fb.hip -- host side of the batched forward-backward (FBFile replacement): chain tables,
// workspace, launches of K1..K4, result collection.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>
#include <thread>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <functional>
#include "internal.h"
#include "hipcheck.h"
#include "kernels.h"

struct DevBuf {
   void *p = nullptr;
   size_t cap = 0;
   bool view = false;            // p points into another allocation (the batch-table arena)
   void set_view(void *q) { if (p && !view) (void)hipFree(p); p = q; cap = 0; view = true; }
   int reserve(size_t bytes)
   {
      if (view) { p = nullptr; view = false; cap = 0; }
      if (bytes <= cap) return HTKAMD_OK;
      if (p) (void)hipFree(p);
      p = nullptr; cap = 0;
      size_t want = bytes + bytes / 8;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) { htkamd_set_error("fb: hipMalloc(%zu bytes): %s", want, hipGetErrorString(e)); return HTKAMD_ENOMEM; }
      cap = want;
      return HTKAMD_OK;
   }
   void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
};

// Small persistent worker pool for the host-side batch preparation (thread creation costs more than a share's work).
struct PrepPool {
   std::vector<std::thread> th;
   std::mutex mu;
   std::condition_variable cvGo, cvDone;
   std::function<void(int)> job;
   int generation = 0, pending = 0, nJobs = 0;
   bool quit = false;
   void start(int n)
   {
      for (int k = 0; k < n; k++)
         th.emplace_back([this, k]() {
            int seen = 0;
            for (;;) {
               std::function<void(int)> f;
               {
                  std::unique_lock<std::mutex> lk(mu);
                  cvGo.wait(lk, [&] { return quit || generation != seen; });
                  if (quit) return;
                  seen = generation;
                  if (k + 1 >= nJobs) { if (--pending == 0) cvDone.notify_one(); continue; }
                  f = job;
               }
               f(k + 1);                                   // share 0 runs on the calling thread
               std::unique_lock<std::mutex> lk(mu);
               if (--pending == 0) cvDone.notify_one();
            }
         });
   }
   // runs f(0..n-1); f(0) on the caller
   void run(int n, const std::function<void(int)> &f)
   {
      if (n <= 1 || th.empty()) { for (int k = 0; k < n; k++) f(k); return; }
      {
         std::unique_lock<std::mutex> lk(mu);
         job = f; nJobs = n; pending = (int)th.size(); generation++;
      }
      cvGo.notify_all();
      f(0);
      std::unique_lock<std::mutex> lk(mu);
      cvDone.wait(lk, [&] { return pending == 0; });
   }
   ~PrepPool()
   {
      { std::unique_lock<std::mutex> lk(mu); quit = true; }
      cvGo.notify_all();
      for (auto &t : th) t.join();
   }
};

// One worker's share of a batch: the tables of a contiguous range of utterances with offsets relative to the share.
struct PrepChunk {
   std::vector<int> mN, mTp, mCell0, mSlot0, mDms, mHmm, mTrans, slotState;
   std::vector<int> slotRange;                // per chain state [first row, last row] of X that Setotprob of the un-pruned pass evaluates (ScoreArgs::slotRange)
   std::vector<int> slotStateU;               // several streams: the (state, stream) element of every chain state, per utterance NSt blocks of nSlots
   std::vector<short> cQ, cI, thrCell, sQ;
   std::vector<ScoreTask> tasks, tasksW;      // scoring tasks in groups of SCORE_TASK_SLOTS chain states (exact kernel) and of SCORE_TASK_SLOTS_WIDE (matrix-core kernels)
   std::vector<int> evLo, evHi, slotModel;
   size_t outp = 0, beta = 0, gam = 0;
   long long frameStates = 0;
   int nCellsMax = 1, QMax = 1, TMax = 1, nThrMax = 64, rc = HTKAMD_OK;
   char err[256] = "";
   void reset()                                          // keeps the vectors' capacity from batch to batch
   {
      mN.clear(); mTp.clear(); mCell0.clear(); mSlot0.clear(); mDms.clear(); mHmm.clear(); mTrans.clear(); slotState.clear(); slotStateU.clear(); slotRange.clear();
      cQ.clear(); cI.clear(); thrCell.clear(); sQ.clear(); tasks.clear(); tasksW.clear();
      outp = beta = gam = 0; frameStates = 0; nCellsMax = QMax = TMax = 1; nThrMax = 64; rc = HTKAMD_OK; err[0] = 0;
   }
};

struct htkamd_fb {
   htkamd_model *m;
   int nUtt;
   int debug;
   int forceGeneral;            // test aid: use the workgroup-per-utterance kernels even when the wave path applies
   int noStatePath;             // test aid: keep utterances off the state-per-lane kernels (fb_state.hip)
   int noLrPath;                // test aid: keep left-to-right chains off their own kernels (fb_lr.hip): they run on fb_state.hip's
   int topoVersion;             // the model's topology version the batch tables were built against
   // host tables of the prepared batch
   std::vector<UttDesc> utt;
   std::vector<int> mN, mTp, mCell0, mSlot0, mDms, mHmm, mTrans, slotState, slotStateU, slotRange;
   std::vector<short> cQ, cI, taperLo, taperHi, thrCell, sQ;
   std::vector<int> wqStart;    // tasksW in eight queues by utterance % 8 (ScoreArgs::qStart): first task of every queue, then the total
   std::vector<int> qBeamNP;    // per frame: the beta beam of the un-pruned pass, lo | hi << 16 (what SetBeta leaves in qLo / qHi when only the taper acts)
   std::vector<ScoreTask> tasks, tasksW;      // scoring tasks in groups of SCORE_TASK_SLOTS chain states (exact kernel) and of SCORE_TASK_SLOTS_WIDE (matrix-core kernels)
   std::vector<size_t> gamOff;
   std::vector<int> gamChunkUtt;
   size_t outpTotal, betaTotal, gamTotal;
   int totalFrames, nCellsMax, QMax, TMax, blockDim;
   long long frameStates;
   const float *dX;
   // device
   DevBuf d_utt, d_mN, d_mTp, d_mCell0, d_mSlot0, d_mDms, d_mHmm, d_mTrans, d_slotState, d_cQ, d_cI, d_taperLo, d_taperHi;
   DevBuf d_tasks, d_tasksW, d_gamOff, d_qLo, d_qHi, d_aLo, d_aHi, d_outp, d_beta, d_gam, d_alpha, d_pr, d_status;
   DevBuf d_betaW;                          // wave path's beta blocks (UttDesc::betaW0)
   DevBuf d_tmE, d_tmMaxP;                  // tied mixtures: the pool's per-frame table (kernels.h FbArgs::tmE)
   std::vector<int> nextSame; DevBuf d_nextSame;   // HTKAMD_COMPAT_STREAM_REVISIT (kernels.h FbArgs::nextSame)
   DevBuf d_slotRange;
   DevBuf d_slotStateU, d_outpU;            // several streams: element of every (stream, chain state), and their scores: stream k of utterance u at outpU[NSt*outp0 + (k*nSlots + slot)*T + t-1]
   DevBuf d_uttList, d_sQ;                  // utterance numbers grouped by class: lane-per-model W = 1 | 2 | 4 | 8 | general | lane-per-state W = 1 | 2 | 4 | 8 | left-to-right W = 1 | 2 | 4 | 8
   std::vector<int> uttList;
   int clsOff[14];                          // class c occupies uttList[clsOff[c] .. clsOff[c+1])
   size_t betaWTotal, alphaWTotal;
   DevBuf d_alphaW, d_qBeam, d_aBeam, d_trPart, d_hits, d_hitCtl;   // left-to-right path (fb_lr.hip)
   DevBuf d_wqStart;                         // ScoreArgs::qStart
   DevBuf d_sink;                            // FbArgs::sink
   DevBuf d_stCnt, d_stBucket;               // FbArgs::stCnt, stBucket (k_mixstate)
   DevBuf d_qBeamNP, d_laneRec;              // ... the host's un-pruned beta beams (a view into the arena); a record per chain state for the sparse statistics
   bool mixStateLast = false;                // the last pass ran k_mixstate (its counters are behind d_stCnt)
   bool mixDeferred = false;                 // htkamd_fb_execute_begin left the state-bucketed statistics to htkamd_fb_execute_mix
   FbArgs *faMix = nullptr;                  // ... with these arguments
   const int *qBeamLast = nullptr;           // the beta beam words the last pass's left-to-right kernels read (d_qBeam or d_qBeamNP)
   bool lastWave;                           // (kept for the tests' introspection) the last execute used no general kernel
   DevBuf d_transOff, d_trOccOff, d_counter, d_thrCell, d_arena, d_gamChunkUtt;
   DevBuf d_rec, d_recSorted, d_recCtl;     // statistics records (kernels.h MixRec)
   int recCapForce;                         // > 0: capacity of the record list (tests: forces the overflow path)
   PrepPool *pool; std::vector<PrepChunk> *chunks;   // host workers and their reusable share buffers
   void *h_arena; size_t h_arenaCap;        // pinned staging copy of the batch tables (one H2D transfer per prepare)
   void *h_res; size_t h_resCap;            // pinned staging copy of the results (one D2H transfer per htkamd_fb_results)
   hipEvent_t evRes; bool resPending;       // htkamd_fb_results_begin: the event behind the queued copy
   bool f16Pass = false;                    // the last pass scored on the fp16 path: its range flag lies behind the status words
   hipEvent_t ev[6], evK[2], evCopy;          // ev: stream intervals (score | beta | alpha | left-to-right statistics | mixture statistics); evK: the scoring dispatch's own start/stop
   hipStream_t resStream;                   // non-blocking stream for fb_results (does not wait for later launches)
   bool evValid, timed, copyPending, scored;
   int evMode = 0, evModeLast = 0;           // htkamd_fb_set_event_mode; the mode of the last htkamd_fb_execute
};

// the reference's second-visit arithmetic is asked for AND can be reached with this set (htkamd_model_set_compat)
static bool compat_revisit(const htkamd_model *m) { return (m->compat & HTKAMD_COMPAT_STREAM_REVISIT) && m->NSt > 1 && m->NSt != 3 && !m->tiedMix; }

extern "C" int htkamd_fb_create(htkamd_model *m, htkamd_fb **out)
{
   if (!m || !out) { htkamd_set_error("fb_create: NULL argument"); return HTKAMD_EINVAL; }
   if (m->maxM > 4096) { htkamd_set_error("fb_create: %d mixture components per state not supported", m->maxM); return HTKAMD_EMODEL; }
   htkamd_fb *fb = new htkamd_fb();
   fb->m = m; fb->nUtt = 0; fb->debug = 0; fb->forceGeneral = 0; fb->evValid = false; fb->timed = false; fb->copyPending = false; fb->scored = false; fb->lastWave = false; fb->betaWTotal = 0; fb->alphaWTotal = 0; fb->noStatePath = 0; fb->noLrPath = 0; fb->recCapForce = 0; for (int c = 0; c < 14; c++) fb->clsOff[c] = 0;
   fb->outpTotal = fb->betaTotal = fb->gamTotal = 0; fb->frameStates = 0; fb->dX = nullptr; fb->h_arena = nullptr; fb->h_arenaCap = 0; fb->h_res = nullptr; fb->h_resCap = 0; fb->evRes = nullptr; fb->resPending = false; fb->pool = nullptr; fb->chunks = nullptr;
   for (int i = 0; i < 6; i++) fb->ev[i] = nullptr;
   fb->evK[0] = fb->evK[1] = fb->evCopy = nullptr; fb->resStream = nullptr;
   fb->evValid = true;                                   // destroy releases whatever has been created (null handles are skipped)
   auto fail = [&](const char *what, hipError_t e) { htkamd_set_error("fb_create: %s: %s", what, hipGetErrorString(e)); htkamd_fb_destroy(fb); return HTKAMD_EHIP; };
   hipError_t e;
   for (int i = 0; i < 6; i++) if ((e = hipEventCreate(&fb->ev[i])) != hipSuccess) return fail("hipEventCreate", e);
   if ((e = hipEventCreate(&fb->evK[0])) != hipSuccess || (e = hipEventCreate(&fb->evK[1])) != hipSuccess) return fail("hipEventCreate", e);
   if ((e = hipEventCreateWithFlags(&fb->evCopy, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
   if ((e = hipStreamCreateWithFlags(&fb->resStream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
   int rc;
   if ((rc = fb->d_counter.reserve(64)) || (rc = fb->d_transOff.reserve(sizeof(int) * (m->nT + 1))) || (rc = fb->d_trOccOff.reserve(sizeof(int) * (m->nT + 1)))) { htkamd_fb_destroy(fb); return rc; }
   if ((e = hipMemcpy(fb->d_transOff.p, m->h_transOff, sizeof(int) * (m->nT + 1), hipMemcpyHostToDevice)) != hipSuccess ||
       (e = hipMemcpy(fb->d_trOccOff.p, m->h_trOccOff, sizeof(int) * (m->nT + 1), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
   *out = fb;
   return HTKAMD_OK;
}

extern "C" void htkamd_fb_destroy(htkamd_fb *fb)
{
   if (!fb) return;
   DevBuf *all[] = {&fb->d_utt, &fb->d_mN, &fb->d_mTp, &fb->d_mCell0, &fb->d_mSlot0, &fb->d_mDms, &fb->d_mHmm, &fb->d_mTrans,
                    &fb->d_slotState, &fb->d_cQ, &fb->d_cI, &fb->d_taperLo, &fb->d_taperHi, &fb->d_tasks, &fb->d_tasksW, &fb->d_gamOff,
                    &fb->d_qLo, &fb->d_qHi, &fb->d_aLo, &fb->d_aHi, &fb->d_outp, &fb->d_beta, &fb->d_gam, &fb->d_alpha,
                    &fb->d_pr, &fb->d_status, &fb->d_betaW, &fb->d_uttList, &fb->d_sQ, &fb->d_transOff, &fb->d_trOccOff, &fb->d_counter, &fb->d_thrCell, &fb->d_arena, &fb->d_gamChunkUtt, &fb->d_rec, &fb->d_recSorted, &fb->d_recCtl, &fb->d_alphaW, &fb->d_qBeam, &fb->d_aBeam, &fb->d_trPart, &fb->d_hits, &fb->d_hitCtl, &fb->d_slotStateU, &fb->d_outpU, &fb->d_tmE, &fb->d_tmMaxP, &fb->d_qBeamNP, &fb->d_slotRange, &fb->d_laneRec, &fb->d_sink, &fb->d_stCnt, &fb->d_stBucket, &fb->d_wqStart};
   for (DevBuf *b : all) b->release();
   if (fb->h_arena) (void)hipHostFree(fb->h_arena);
   if (fb->h_res) (void)hipHostFree(fb->h_res);
   delete fb->pool; delete fb->chunks; delete fb->faMix;
   if (fb->evValid) {
      for (int i = 0; i < 6; i++) if (fb->ev[i]) (void)hipEventDestroy(fb->ev[i]);
      if (fb->evCopy) (void)hipEventDestroy(fb->evCopy);
      if (fb->evRes) (void)hipEventDestroy(fb->evRes);
      if (fb->evK[0]) (void)hipEventDestroy(fb->evK[0]);
      if (fb->evK[1]) (void)hipEventDestroy(fb->evK[1]);
      if (fb->resStream) (void)hipStreamDestroy(fb->resStream);
   }
   delete fb;
}

extern "C" int htkamd_fb_set_debug(htkamd_fb *fb, int on)
{
   if (!fb) { htkamd_set_error("fb_set_debug: NULL"); return HTKAMD_EINVAL; }
   fb->debug = on & 1;
   fb->forceGeneral = (on & 2) ? 1 : 0;
   fb->noStatePath = (on & 4) ? 1 : 0;
   fb->recCapForce = (on & 8) ? 128 : ((on & 16) ? -1 : 0);          // 8: a 128-record list (overflow path), 16: no record list (direct atomics)
   fb->noLrPath = (on & 32) ? 1 : 0;
   return HTKAMD_OK;
}

template <typename T> static int upload(DevBuf &b, const std::vector<T> &v, hipStream_t s)
{
   int rc = b.reserve(sizeof(T) * (v.size() ? v.size() : 1));
   if (rc) return rc;
   if (!v.empty()) HIPCHECK(hipMemcpyAsync(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s));
   return HTKAMD_OK;
}

// CreateInsts (HFB.c:508-574) + SetBeamTaper (HFB.c:1116-1145) + the scoring tasks of utterance u, appended to chunk C.
static int prep_utterance(htkamd_fb *fb, const htkamd_batch_desc *b, int u, PrepChunk &C)
{
   const htkamd_model *m = fb->m;
   std::vector<int> &evLo = C.evLo, &evHi = C.evHi, &slotModel = C.slotModel;
      UttDesc &d = fb->utt[u];
      const int T = b->frameOff[u + 1] - b->frameOff[u], Q = b->labOff[u + 1] - b->labOff[u];
      const int *labs = b->labs + b->labOff[u];
      d.T = T; d.Q = Q; d.frame0 = b->frameOff[u];
      d.q0 = (int)C.mN.size(); d.cell0 = (int)C.cQ.size(); d.slot0 = (int)C.slotState.size();
      d.status = HTKAMD_UTT_OK; d.nEval = 0;
      d.outp0 = C.outp; d.beta0 = C.beta; d.gam0 = C.gam;
      
      d.pad2 = 3;                                   // bit 0: no tee model in the chain, bit 1: every model left-to-right without skips (read and cleared by htkamd_fb_prepare's class loop)
      if (T <= 0 || Q <= 0) { d.status = HTKAMD_UTT_SKIPPED; d.nCells = d.nSlots = 0; d.thr0 = (int)C.thrCell.size(); d.nThr = 0; return HTKAMD_OK; }
      int nCells = 0, nSlots = 0, qt = 0, prevDm = 1;
      for (int q = 1; q <= Q; q++) {
         const int h = labs[q - 1];
         if (h < 0 || h >= m->H) { snprintf(C.err, sizeof(C.err), "fb_prepare: utterance %d label %d: HMM index %d out of range", u, q, h); return HTKAMD_EINVAL; }
         const int ti = m->h_hmmTrans[h], N = m->h_transN[ti], dm = m->h_minDur[ti];
         C.mN.push_back(N); C.mTp.push_back(m->h_transOff[ti]); C.mCell0.push_back(nCells); C.mSlot0.push_back(nSlots);
         C.mDms.push_back(dm); C.mHmm.push_back(h); C.mTrans.push_back(ti);
         for (int i = 1; i <= N; i++) { C.cQ.push_back((short)q); C.cI.push_back((short)i); }
         for (int j = 2; j < N; j++) { C.slotState.push_back(m->h_hmmState[m->h_hmmStateOff[h] + (j - 2) * m->NSt]); C.sQ.push_back((short)q); C.slotRange.push_back(0); C.slotRange.push_back(-1); }      // several streams: the state's first element
         nCells += N; nSlots += N - 2; qt += dm;
         if (dm == 0) d.pad2 &= ~1;
         if (m->h_transLR[ti] != 1) d.pad2 &= ~2;
         if (q > 1 && dm == 0 && prevDm == 0) d.status = HTKAMD_UTT_ETEE;      // successive tee models (HFB.c:557)
         prevDm = dm;
      }
      if (C.mDms[d.q0] == 0 || C.mDms[d.q0 + Q - 1] == 0) d.status = HTKAMD_UTT_ETEE;   // HFB.c:564
      if (d.status == HTKAMD_UTT_OK && qt > T) d.status = HTKAMD_UTT_SKIPPED;                 // HFB.c:1339
      d.nCells = nCells; d.nSlots = nSlots;
      if (m->NSt > 1) for (int ks = 0; ks < m->NSt; ks++) for (int k = 0; k < nSlots; k++) C.slotStateU.push_back(C.slotState[d.slot0 + k] + ks);
      {  // thread map: entry cells | emitting cells | exit cells, each group padded to a multiple of 64
         d.thr0 = (int)C.thrCell.size();
         for (int role = 0; role < 3; role++) {
            for (int q = 1; q <= Q; q++) {
               const int c0 = C.mCell0[d.q0 + q - 1], N = C.mN[d.q0 + q - 1];
               if (role == 0) C.thrCell.push_back((short)c0);
               else if (role == 2) C.thrCell.push_back((short)(c0 + N - 1));
               else for (int i = 2; i < N; i++) C.thrCell.push_back((short)(c0 + i - 1));
            }
            while ((C.thrCell.size() - d.thr0) % 64) C.thrCell.push_back((short)-1);
         }
         d.nThr = (int)C.thrCell.size() - d.thr0;
      }
      // chains of up to 512 models of up to 5 states run on 1..8 wavefronts (fb_wave.hip); the others on the general kernels: one
      // thread per model state, everything of a frame in LDS
      const bool waveOk = fb->m->maxN <= 5 && !fb->forceGeneral && !compat_revisit(fb->m) && Q <= 512;
      if (!waveOk) {
         if (Q > 32000 || d.nThr > 1024) {
            snprintf(C.err, sizeof(C.err), "fb_prepare: utterance %d has %d models / %d model states; the device path handles chains of up to 512 models of "
                     "up to 5 states, or up to 1024 model states per utterance", u, Q, nCells);
            return HTKAMD_EINVAL;
         }
         if (d.nThr > C.nThrMax) C.nThrMax = d.nThr;
         if (nCells > C.nCellsMax) C.nCellsMax = nCells;
         if (Q > C.QMax) C.QMax = Q;
      }
      if (T > C.TMax) C.TMax = T;
      C.outp += (size_t)T * nSlots; C.beta += (size_t)T * nCells; C.gam += (size_t)T * nSlots;
      if (d.status != HTKAMD_UTT_OK) {
         // (the per-frame tables are filled only when the batch's size changes: an utterance that stops here must not keep the words of another
         //  batch's utterance at its frames -- htkamd_fb_get_trellis reads qBeamNP for the left-to-right class)
         for (int t = 0; t < T; t++) { fb->taperLo[d.frame0 + t] = 0; fb->taperHi[d.frame0 + t] = 0; fb->qBeamNP[d.frame0 + t] = 1; }
         return HTKAMD_OK;
      }
      // SetBeamTaper
      short *lo = fb->taperLo.data() + d.frame0 - 1, *hi = fb->taperHi.data() + d.frame0 - 1;
      const int *dms = C.mDms.data() + d.q0 - 1;                         // 1-based q
      {
         int q = 1, dq = dms[q], i = 0;
         for (int t = 1; t <= T; t++) {
            while (i == dq) { i = 0; if (q < Q) { q++; dq = dms[q]; } else dq = -1; }
            hi[t] = (short)q; i++;
         }
         q = Q; dq = dms[q]; i = 0;
         for (int t = T; t >= 1; t--) {
            while (i == dq) { i = 0; if (q > 1) { q--; dq = dms[q]; } else dq = -1; }
            lo[t] = (short)q; i++;
         }
      }
      // Setotprob ranges of the un-pruned pass (HFB.c:1177,1215 + 1014): models evLo[t]..evHi[t] are scored at t.
      // They bound what any pass can touch (pruning only narrows them) and give the metric's unit count.
      evLo.assign(T + 2, 0); evHi.assign(T + 2, 0);
      {
         const long long before = C.frameStates;
         int qHiN = Q, qLoN = lo[T];
         int *npw = fb->qBeamNP.data() + d.frame0 - 1;                      // 1-based t
         npw[T] = qLoN | (qHiN << 16);
         const int *msl = C.mSlot0.data() + d.q0 - 1;
         auto slotsIn = [&](int a, int z) { return (z < Q ? msl[z + 1] : nSlots) - msl[a]; };
         evLo[T] = qLoN > 1 ? qLoN - 1 : 1; evHi[T] = Q;
         C.frameStates += slotsIn(evLo[T], Q);
         for (int t = T - 1; t >= 1; t--) {
            const int startq = qHiN;
            int endq = (qLoN == 1) ? 1 : ((lo[t] >= qLoN) ? lo[t] : qLoN - 1);
            while (endq > 1 && dms[endq - 1] == 0) endq--;
            evLo[t] = endq > 1 ? endq - 1 : 1; evHi[t] = startq;
            C.frameStates += slotsIn(evLo[t], startq);
            qHiN = (hi[t] < startq) ? hi[t] : startq; qLoN = endq;
            npw[t] = qLoN | (qHiN << 16);
         }
         d.nEval = (int)(C.frameStates - before);
      }
      {  // per chain state the frames in which its model lies inside [evLo, evHi] (both rise with t): first t with evHi[t] >= q .. last t with evLo[t] <= q
         int *rng = C.slotRange.data() + 2 * (size_t)d.slot0;
         int t = 1;
         for (int q = 1; q <= Q; q++) {
            while (t <= T && evHi[t] < q) t++;
            const int s0 = C.mSlot0[d.q0 + q - 1], n = C.mN[d.q0 + q - 1] - 2;
            for (int j = 0; j < n; j++) rng[2 * (s0 + j)] = d.frame0 + t - 1;
         }
         t = T;
         for (int q = Q; q >= 1; q--) {
            while (t >= 1 && evLo[t] > q) t--;
            const int s0 = C.mSlot0[d.q0 + q - 1], n = C.mN[d.q0 + q - 1] - 2;
            for (int j = 0; j < n; j++) rng[2 * (s0 + j) + 1] = d.frame0 + t - 1;
         }
      }
      // scoring tasks: chunks of chain states x tiles of the frames in which the chunk can be in the beam
      {
         const short *cq = C.cQ.data() + d.cell0;
         (void)cq;
         // model of every slot
         slotModel.resize(nSlots);
         for (int q = 1; q <= Q; q++) {
            const int s0 = C.mSlot0[d.q0 + q - 1], n = C.mN[d.q0 + q - 1] - 2;
            for (int j = 0; j < n; j++) slotModel[s0 + j] = q;
         }
         // two partitions of the same rectangle set: the matrix-core kernels build their B operand (the task's 128 frames, split into
         // bf16 pieces) once per task, so they get four times as many states per task (measured: 1.32 -> 1.21 ms at the bench workload;
         // the exact kernel is 10 % slower on the wide tasks)
         // several streams: every stream of a chain state is a row of its own in the score block (NSt blocks of nSlots rows per
         // utterance, k_combine_streams sums them into the state's row of outp)
         const int NSt = m->NSt;
         for (int wide = 0; wide < 2; wide++) {
         const int GS = wide ? SCORE_TASK_SLOTS_WIDE : SCORE_TASK_SLOTS_EXACT;
         std::vector<ScoreTask> &dst = wide ? C.tasksW : C.tasks;
         for (int ks = 0; ks < NSt; ks++)
         for (int k0 = 0; k0 < nSlots; k0 += GS) {
            const int k1 = (k0 + GS < nSlots) ? k0 + GS : nSlots;
            const int qa = slotModel[k0], qb = slotModel[k1 - 1];
            int tmin = 1, tmax = T;
            while (tmin <= T && evHi[tmin] < qa) tmin++;
            while (tmax >= 1 && evLo[tmax] > qb) tmax--;
            const int TF = wide ? B16_TASK_FRAMES : SCORE_TILE_FRAMES;
            for (int t0 = tmin - 1; t0 < tmax; t0 += TF) {
               ScoreTask tk;
               tk.frame0 = d.frame0 + t0;
               tk.nFrames = (tmax - t0 < TF) ? tmax - t0 : TF;
               tk.slot0 = d.slot0 * NSt + ks * nSlots + k0;
               tk.nSlots = k1 - k0;
               tk.outSlot0 = ks * nSlots + k0; tk.ldo = T;
               tk.outBase = d.outp0 * NSt + (size_t)t0;
               dst.push_back(tk);
            }
         }
         }
      }
   return HTKAMD_OK;
}

// CreateInsts (HFB.c:508-574) + SetBeamTaper (HFB.c:1116-1145) for every utterance of the batch,
// and the flat tables the kernels index.  The utterances are independent: host threads take contiguous shares and the
// shares are concatenated with their offsets rebased (the reference does this work inside its per-file loop).
extern "C" int htkamd_fb_prepare(htkamd_fb *fb, const htkamd_batch_desc *b, void *stream)
{
   if (!fb || !b || b->nUtt < 0 || (b->nUtt > 0 && (!b->dX || !b->frameOff || !b->labOff || !b->labs))) {
      htkamd_set_error("fb_prepare: bad argument"); return HTKAMD_EINVAL;
   }
   hipStream_t s = (hipStream_t)stream;
   const int U = b->nUtt;
   // check the batch description before any state of `fb` is touched: a failing call leaves the previous batch as it was
   if (U > 0) {
      if (b->frameOff[0] < 0 || b->labOff[0] < 0) { htkamd_set_error("fb_prepare: negative first offset"); return HTKAMD_EINVAL; }
      for (int u = 0; u < U; u++)
         if (b->frameOff[u + 1] < b->frameOff[u] || b->labOff[u + 1] < b->labOff[u]) {
            htkamd_set_error("fb_prepare: frameOff / labOff must be non-decreasing (utterance %d)", u); return HTKAMD_EINVAL;
         }
   }
   // from here on a failure invalidates the context: execute / results / get_trellis then see an empty batch, never a mixture of two
   struct Invalidate { htkamd_fb *f; bool ok; ~Invalidate() { if (!ok) { f->nUtt = 0; f->timed = false; } } } guard{fb, false};
   static const bool timing = getenv("HTKAMD_PREP_TIMING") != nullptr;
   auto tp0 = std::chrono::steady_clock::now();
   auto lap = [&](const char *what) { if (!timing) return; auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "  prepare %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tp0).count()); tp0 = t; };
   fb->nUtt = U; fb->dX = b->dX; fb->topoVersion = fb->m->topoVersion;
   fb->mixDeferred = false;                              // (statistics left waiting by htkamd_fb_execute_begin belong to the batch before)
   fb->utt.assign(U, UttDesc());
   fb->totalFrames = U ? b->frameOff[U] : 0;
   // (filled only when the batch's size changes: the workers write every frame of every utterance that a kernel will look at -- utterances that
   //  fail CreateInsts keep whatever the arrays held, and no kernel reads their frames -- and 5 MB of fills per prepare were 0.15 ms of the host's loop)
   if (fb->taperLo.size() != (size_t)fb->totalFrames) { fb->taperLo.assign(fb->totalFrames, 0); fb->taperHi.assign(fb->totalFrames, 0); }
   if (fb->qBeamNP.size() != (size_t)fb->totalFrames + 1) fb->qBeamNP.assign((size_t)fb->totalFrames + 1, 1);
   fb->gamOff.assign(U + 1, 0);
   if (!fb->pool) {
      int hw = (int)std::thread::hardware_concurrency();
      if (hw > 16) hw = 16;
      if (hw < 1) hw = 1;
      fb->pool = new PrepPool();
      fb->pool->start(hw - 1);
      fb->chunks = new std::vector<PrepChunk>(hw);
   }
   int nW = (int)fb->chunks->size();
   if (nW > U / 32) nW = U / 32;
   if (nW < 1) nW = 1;
   std::vector<PrepChunk> &chunks = *fb->chunks;
   fb->pool->run(nW, [&](int k) {
      PrepChunk &C = chunks[k];
      C.reset();
      const int u0 = (int)((long long)U * k / nW), u1 = (int)((long long)U * (k + 1) / nW);
      for (int u = u0; u < u1; u++)
         if ((C.rc = prep_utterance(fb, b, u, C))) return;
   });
   lap("workers");
   for (int k = 0; k < nW; k++) if (chunks[k].rc) { htkamd_set_error("%s", chunks[k].err); return chunks[k].rc; }
   // concatenate the shares, rebasing their offsets
   // (the tables are NOT cleared first: resize() of an emptied vector writes zeros over every element the workers are about to fill)
   fb->nCellsMax = 1; fb->QMax = 1; fb->TMax = 1; fb->frameStates = 0;
   int nThrMax = 64;
   size_t outp = 0, beta = 0, gam = 0;
   {
      // where every share's part begins in the batch's tables, then every worker rebases and copies its own share (the copies were a
      // third of the call, one thread's)
      struct Base { size_t q, cell, slot, slotU, thr, tasks, tasksW, outp, beta, gam; };
      std::vector<Base> base((size_t)nW + 1);
      Base z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < nW; k++) {
         const PrepChunk &C = chunks[k];
         base[k] = z;
         z.q += C.mN.size(); z.cell += C.cQ.size(); z.slot += C.slotState.size(); z.slotU += C.slotStateU.size(); z.thr += C.thrCell.size();
         z.tasks += C.tasks.size(); z.tasksW += C.tasksW.size(); z.outp += C.outp; z.beta += C.beta; z.gam += C.gam;
         fb->frameStates += C.frameStates;
         if (C.nCellsMax > fb->nCellsMax) fb->nCellsMax = C.nCellsMax;
         if (C.QMax > fb->QMax) fb->QMax = C.QMax;
         if (C.TMax > fb->TMax) fb->TMax = C.TMax;
         if (C.nThrMax > nThrMax) nThrMax = C.nThrMax;
      }
      base[nW] = z;
      outp = z.outp; beta = z.beta; gam = z.gam;
      fb->mN.resize(z.q); fb->mTp.resize(z.q); fb->mCell0.resize(z.q); fb->mSlot0.resize(z.q); fb->mDms.resize(z.q); fb->mHmm.resize(z.q); fb->mTrans.resize(z.q);
      fb->slotState.resize(z.slot); fb->slotRange.resize(2 * z.slot); fb->sQ.resize(z.slot); fb->slotStateU.resize(z.slotU); fb->cQ.resize(z.cell); fb->cI.resize(z.cell); fb->thrCell.resize(z.thr);
      fb->tasks.resize(z.tasks); fb->tasksW.resize(z.tasksW);
      const int NSt = fb->m->NSt;
      fb->pool->run(nW, [&](int k) {
         PrepChunk &C = chunks[k];
         const Base &B = base[k];
         const int u0 = (int)((long long)U * k / nW), u1 = (int)((long long)U * (k + 1) / nW);
         for (int u = u0; u < u1; u++) {
            UttDesc &d = fb->utt[u];
            d.q0 += (int)B.q; d.cell0 += (int)B.cell; d.slot0 += (int)B.slot; d.thr0 += (int)B.thr; d.outp0 += B.outp; d.beta0 += B.beta; d.gam0 += B.gam;
            fb->gamOff[u] = d.gam0;
         }
         for (ScoreTask &tk : C.tasks) { tk.slot0 += (int)B.slot * NSt; tk.outBase += B.outp * NSt; }
         for (ScoreTask &tk : C.tasksW) { tk.slot0 += (int)B.slot * NSt; tk.outBase += B.outp * NSt; }
         auto put = [](auto &dst, size_t at, const auto &src) { if (!src.empty()) memcpy(dst.data() + at, src.data(), sizeof(src[0]) * src.size()); };
         put(fb->mN, B.q, C.mN); put(fb->mTp, B.q, C.mTp); put(fb->mCell0, B.q, C.mCell0); put(fb->mSlot0, B.q, C.mSlot0); put(fb->mDms, B.q, C.mDms); put(fb->mHmm, B.q, C.mHmm);
         put(fb->mTrans, B.q, C.mTrans); put(fb->slotState, B.slot, C.slotState); put(fb->slotRange, 2 * B.slot, C.slotRange); put(fb->sQ, B.slot, C.sQ); put(fb->slotStateU, B.slotU, C.slotStateU);
         put(fb->cQ, B.cell, C.cQ); put(fb->cI, B.cell, C.cI); put(fb->thrCell, B.thr, C.thrCell); put(fb->tasks, B.tasks, C.tasks); put(fb->tasksW, B.tasksW, C.tasksW);
      });
   }
   fb->gamOff[U] = gam;
   lap("merge:copy");
   {  // the wide tasks in eight queues, utterance u in queue u % 8 (the tasks lie in utterance order: one walk finds their utterances)
      std::vector<ScoreTask> q[8];
      int u = 0;
      for (const ScoreTask &tk : fb->tasksW) {
         while (u + 1 < U && tk.frame0 >= b->frameOff[u + 1]) u++;
         q[u & 7].push_back(tk);
      }
      fb->wqStart.assign(9, 0);
      size_t at = 0;
      for (int k = 0; k < 8; k++) {
         fb->wqStart[k] = (int)at;
         if (!q[k].empty()) memcpy(fb->tasksW.data() + at, q[k].data(), sizeof(ScoreTask) * q[k].size());
         at += q[k].size();
      }
      fb->wqStart[8] = (int)at;
   }
   lap("merge:queues");
   fb->outpTotal = outp; fb->betaTotal = beta; fb->gamTotal = gam;
   fb->blockDim = nThrMax;
   {  // classes: chains of <= 64 / 128 / 256 models of <= 5 states go to the wave kernels with 1 / 2 / 4 wavefronts, the rest to
      // the general workgroup-per-utterance kernels
      std::vector<int> cls[13];
      size_t bw = 0, aw = 0;
      for (int u = 0; u < U; u++) {
         UttDesc &d = fb->utt[u];
         // a lane per chain state (fb_state.hip) where the chain has no tee model and at most 512 emitting states; else a lane per model
         // (fb_wave.hip, chains of up to 512 models); else the general workgroup-per-utterance kernels
         const bool noTee = (d.status == HTKAMD_UTT_OK || d.status == HTKAMD_UTT_SKIPPED) && (d.pad2 & 1);      // (the models were looked at by the workers)
         const bool small = fb->m->maxN <= 5 && !fb->forceGeneral && !compat_revisit(fb->m);
         int W = 0, kind = 0;
         if (small && !fb->noStatePath && noTee && d.nSlots >= 1 && d.nSlots <= 512) {
            kind = 1; W = d.nSlots <= 64 ? 1 : d.nSlots <= 128 ? 2 : d.nSlots <= 256 ? 4 : 8;
            // every model left-to-right without skips: the kernels of fb_lr.hip (no statistics in the alpha chain, no entry-state columns)
            const bool lr = !fb->noLrPath && (d.pad2 & 2);      // (several streams / tied mixtures too since round 4: their pairs go to the list with the chain state's slot, k_mixhits_streams)
            if (lr) kind = 2;
         }
         else if (small) W = d.Q <= 64 ? 1 : d.Q <= 128 ? 2 : d.Q <= 256 ? 4 : d.Q <= 512 ? 8 : 0;
         d.W = W; d.pad = kind; d.betaW0 = bw; d.alphaW0 = aw; d.QP = (d.Q + 7) & ~7; d.pad2 = 0;
         bw += kind == 2 ? (size_t)d.T * 64 * W : kind ? (size_t)d.T * 2 * 64 * W : (size_t)d.T * 5 * 64 * W;
         if (kind == 2) aw += (size_t)d.T * (64 * W + d.QP);
         const int wc = W == 1 ? 0 : W == 2 ? 1 : W == 4 ? 2 : W == 8 ? 3 : 4;
         cls[(kind == 2 && W) ? 9 + wc : (kind && W) ? 5 + wc : wc].push_back(u);
      }
      fb->betaWTotal = bw; fb->alphaWTotal = aw;
      // within a class the longest utterances are dispatched first (their recursions are the critical path when the batch is larger
      // than the wavefront slots of the machine)
      for (int c = 0; c < 13; c++)
         std::stable_sort(cls[c].begin(), cls[c].end(), [&](int x, int y) { return fb->utt[x].T > fb->utt[y].T; });
      fb->uttList.clear(); fb->clsOff[0] = 0;
      for (int c = 0; c < 13; c++) { fb->uttList.insert(fb->uttList.end(), cls[c].begin(), cls[c].end()); fb->clsOff[c + 1] = (int)fb->uttList.size(); }
      if (fb->uttList.empty()) fb->uttList.push_back(0);
   }
   {  // utterance of every 512th seed (the scan chunks of k_mixstats and its several-stream / tied-mixture forms: the utterances that are
      // NOT on the left-to-right path -- a batch without such utterances does not build or upload the table, 115 000 entries at the bench's)
      const bool anyGeneral = fb->clsOff[9] > 0;
      const size_t nChunk = anyGeneral ? (gam + 511) / 512 : 0;
      fb->gamChunkUtt.assign(nChunk ? nChunk : 1, 0);
      int u = 0;
      for (size_t c = 0; c < nChunk; c++) {
         while (u + 1 < U && fb->gamOff[u + 1] <= c * 512) u++;
         fb->gamChunkUtt[c] = u;
      }
   }

   lap("merge:classes");
   fb->nextSame.clear();
   if (compat_revisit(fb->m)) {
      // per utterance: the chain states of one tied state sorted by (model ascending, state descending); a state's successor in that
      // list is the one Setotprob visited last before it among those that can be in the same call (it walks the models right to left,
      // a model's states left to right: HFB.c:1015-1024)
      fb->nextSame.assign(fb->slotState.size(), -1);
      std::vector<int> ord;
      for (int u = 0; u < U; u++) {
         const UttDesc &d = fb->utt[u];
         ord.resize(d.nSlots);
         for (int k = 0; k < d.nSlots; k++) ord[k] = k;
         std::sort(ord.begin(), ord.end(), [&](int x, int y) {
            const int ex = fb->slotState[d.slot0 + x], ey = fb->slotState[d.slot0 + y];
            if (ex != ey) return ex < ey;
            const int qx = fb->sQ[d.slot0 + x], qy = fb->sQ[d.slot0 + y];
            if (qx != qy) return qx < qy;
            return x > y;                                    // same model: slots ascend with the state
         });
         for (int k = 0; k + 1 < d.nSlots; k++)
            if (fb->slotState[d.slot0 + ord[k]] == fb->slotState[d.slot0 + ord[k + 1]]) fb->nextSame[d.slot0 + ord[k]] = ord[k + 1];
      }
   }
   lap("merge");
   int rc;
   {
      // all tables through one pinned staging buffer and ONE host-to-device copy
      struct Part { DevBuf *dst; const void *src; size_t bytes, off; };
      Part parts[] = {
         {&fb->d_utt, fb->utt.data(), sizeof(UttDesc) * fb->utt.size(), 0}, {&fb->d_mN, fb->mN.data(), sizeof(int) * fb->mN.size(), 0},
         {&fb->d_mTp, fb->mTp.data(), sizeof(int) * fb->mTp.size(), 0}, {&fb->d_mCell0, fb->mCell0.data(), sizeof(int) * fb->mCell0.size(), 0},
         {&fb->d_mSlot0, fb->mSlot0.data(), sizeof(int) * fb->mSlot0.size(), 0}, {&fb->d_mDms, fb->mDms.data(), sizeof(int) * fb->mDms.size(), 0},
         {&fb->d_mHmm, fb->mHmm.data(), sizeof(int) * fb->mHmm.size(), 0}, {&fb->d_mTrans, fb->mTrans.data(), sizeof(int) * fb->mTrans.size(), 0},
         {&fb->d_slotState, fb->slotState.data(), sizeof(int) * fb->slotState.size(), 0}, {&fb->d_cQ, fb->cQ.data(), sizeof(short) * fb->cQ.size(), 0},
         {&fb->d_thrCell, fb->thrCell.data(), sizeof(short) * fb->thrCell.size(), 0}, {&fb->d_cI, fb->cI.data(), sizeof(short) * fb->cI.size(), 0},
         {&fb->d_taperLo, fb->taperLo.data(), sizeof(short) * fb->taperLo.size(), 0}, {&fb->d_taperHi, fb->taperHi.data(), sizeof(short) * fb->taperHi.size(), 0},
         {&fb->d_tasks, fb->tasks.data(), sizeof(ScoreTask) * fb->tasks.size(), 0}, {&fb->d_tasksW, fb->tasksW.data(), sizeof(ScoreTask) * fb->tasksW.size(), 0}, {&fb->d_gamOff, fb->gamOff.data(), sizeof(size_t) * fb->gamOff.size(), 0},
         {&fb->d_gamChunkUtt, fb->gamChunkUtt.data(), sizeof(int) * fb->gamChunkUtt.size(), 0},
         {&fb->d_uttList, fb->uttList.data(), sizeof(int) * fb->uttList.size(), 0}, {&fb->d_sQ, fb->sQ.data(), sizeof(short) * fb->sQ.size(), 0},
         {&fb->d_slotStateU, fb->slotStateU.data(), sizeof(int) * fb->slotStateU.size(), 0},
         {&fb->d_nextSame, fb->nextSame.data(), sizeof(int) * fb->nextSame.size(), 0},
         {&fb->d_qBeamNP, fb->qBeamNP.data(), sizeof(int) * fb->qBeamNP.size(), 0}, {&fb->d_slotRange, fb->slotRange.data(), sizeof(int) * fb->slotRange.size(), 0},
         {&fb->d_wqStart, fb->wqStart.data(), sizeof(int) * fb->wqStart.size(), 0}};
      size_t total = 0;
      for (Part &q : parts) { q.off = total; total += (q.bytes + 255) & ~(size_t)255; }
      if (total == 0) total = 256;
      if (total > fb->h_arenaCap) {
         if (fb->h_arena) (void)hipHostFree(fb->h_arena);
         fb->h_arena = nullptr; fb->h_arenaCap = 0;
         const size_t want = total + total / 4;
         HIPCHECK(hipHostMalloc(&fb->h_arena, want, hipHostMallocDefault));
         fb->h_arenaCap = want;
      }
      if (fb->copyPending) { HIPCHECK(hipEventSynchronize(fb->evCopy)); fb->copyPending = false; }   // previous batch still in flight
      for (Part &q : parts) q.dst->release();
      if ((rc = fb->d_arena.reserve(total))) return rc;
      {  // the staging copy by the workers too: parts cut into 256 KB pieces
         struct Piece { char *dst; const char *src; size_t n; };
         std::vector<Piece> pieces;
         for (Part &q : parts) {
            for (size_t o = 0; o < q.bytes; o += (size_t)256 << 10)
               pieces.push_back({(char *)fb->h_arena + q.off + o, (const char *)q.src + o, (q.bytes - o < ((size_t)256 << 10)) ? q.bytes - o : ((size_t)256 << 10)});
            q.dst->set_view((char *)fb->d_arena.p + q.off);
         }
         const int nP = (int)pieces.size(), nT = nW < 8 ? nW : 8;
         fb->pool->run(nT, [&](int k) { for (int i = k; i < nP; i += nT) memcpy(pieces[i].dst, pieces[i].src, pieces[i].n); });
      }
      HIPCHECK(hipMemcpyAsync(fb->d_arena.p, fb->h_arena, total, hipMemcpyHostToDevice, s));
      HIPCHECK(hipEventRecord(fb->evCopy, s));
      fb->copyPending = true;
   }
   lap("stage+copy");
   // the wave-per-utterance kernels keep beta in their own state-major block (d_betaW, reserved in execute)
   const bool wavePathPrep = fb->clsOff[5] == fb->clsOff[4];          // no utterance needs the general kernels
   const size_t nf = fb->totalFrames ? fb->totalFrames : 1;
   if ((rc = fb->d_qLo.reserve(sizeof(short) * nf)) || (rc = fb->d_qHi.reserve(sizeof(short) * nf)) ||
       (rc = fb->d_aLo.reserve(sizeof(short) * nf)) || (rc = fb->d_aHi.reserve(sizeof(short) * nf)) ||
       (rc = fb->d_outp.reserve(sizeof(float) * (outp + 16))) || (rc = fb->d_beta.reserve(sizeof(double) * ((beta && !wavePathPrep) ? beta : 1))) ||
       (rc = fb->d_gam.reserve(sizeof(double) * ((gam && fb->clsOff[9] > 0) ? gam : 1))) || (rc = fb->d_pr.reserve((sizeof(double) + sizeof(int)) * (size_t)(U ? U : 1) + 2 * sizeof(int))) ||      /* log probabilities, then the status words, then the fp16 scoring path's task counter and range flag: ONE copy brings all back */
       (fb->m->NSt > 1 && (rc = fb->d_outpU.reserve(sizeof(float) * (outp * fb->m->NSt + 16)))) ||
       (fb->m->tiedMix && ((rc = fb->d_tmE.reserve(sizeof(float) * (nf * fb->m->tmPool + 16))) || (rc = fb->d_tmMaxP.reserve(sizeof(float) * (nf * fb->m->NSt + 16))))))
      return rc;
   if (fb->debug && (rc = fb->d_alpha.reserve(sizeof(double) * (beta ? beta : 1)))) return rc;
   lap("reserve");
   // no synchronisation here: the copy is stream-ordered before the kernels of execute, and the staging buffer is
   // guarded by evCopy against being refilled while the copy is still in flight
   guard.ok = true;
   return HTKAMD_OK;
}

extern "C" long long htkamd_fb_frame_states(const htkamd_fb *fb) { return fb ? fb->frameStates : 0; }

extern "C" int htkamd_fb_score_work(const htkamd_fb *fb, long long out[3])
{
   if (!fb || !out) { htkamd_set_error("fb_score_work: NULL argument"); return HTKAMD_EINVAL; }
   long long skip = 0, all = 0;
   const bool ranges = fb->m->NSt == 1 && fb->slotRange.size() == 2 * fb->slotState.size();
   for (const ScoreTask &tk : fb->tasksW) {
      const int nPairs = (tk.nSlots + 1) >> 1;
      for (int w = 0; w < B16_TASK_FRAMES / 32; w++) {
         if (32 * w >= tk.nFrames) break;
         all += nPairs;
         if (!ranges) { skip += nPairs; continue; }
         const int r0 = tk.frame0 + 32 * w, r1 = r0 + 31;
         for (int j = 0; j < nPairs; j++) {
            const int k0 = tk.slot0 + 2 * j, k1 = (2 * j + 1 < tk.nSlots) ? k0 + 1 : k0;
            const int lo = std::min(fb->slotRange[2 * k0], fb->slotRange[2 * k1]), hi = std::max(fb->slotRange[2 * k0 + 1], fb->slotRange[2 * k1 + 1]);
            if (lo <= r1 && hi >= r0) skip++;
         }
      }
   }
   out[0] = skip; out[1] = all; out[2] = (fb->frameStates + 63) / 64;
   return HTKAMD_OK;
}

extern "C" int htkamd_fb_prepared_current(const htkamd_fb *fb) { return fb && fb->topoVersion == fb->m->topoVersion; }

static int fb_execute_impl(htkamd_fb *fb, const htkamd_fb_config *cfg, htkamd_accs *accs, void *stream, bool defer);

extern "C" int htkamd_fb_execute(htkamd_fb *fb, const htkamd_fb_config *cfg, htkamd_accs *accs, void *stream)
{
   return fb_execute_impl(fb, cfg, accs, stream, false);
}

// The pass in two phases, for hosts that send a range of states' statistics on their way while the next range is still being summed (the
// accumulator exchange of a multi-GPU HERest: HERest.c:514-557 dumps and merges whole files behind the pass).  _begin runs everything but
// the state-bucketed mixture statistics and says whether such statistics are waiting (*deferred; 0: the pass was of another kind and is
// complete); _mix then takes tied states [state0, state1) -- every state once per pass, in any order.  Behind _mix of a range the
// accumulators of its states' Gaussians, components and the states' own counts are final on this rank (htkamd_accs_state_ranges);
// everything no state owns -- transitions, counters -- is final behind _begin.
extern "C" int htkamd_fb_execute_begin(htkamd_fb *fb, const htkamd_fb_config *cfg, htkamd_accs *accs, void *stream, int *deferred)
{
   if (deferred) *deferred = 0;
   const int rc = fb_execute_impl(fb, cfg, accs, stream, true);
   if (rc == HTKAMD_OK && deferred && fb && fb->nUtt > 0) *deferred = fb->mixDeferred ? 1 : 0;
   return rc;
}

extern "C" int htkamd_fb_execute_mix(htkamd_fb *fb, int state0, int state1, void *stream)
{
   if (!fb) { htkamd_set_error("fb_execute_mix: NULL"); return HTKAMD_EINVAL; }
   if (fb->nUtt == 0 || !fb->mixDeferred) return HTKAMD_OK;
   const int rc = htkamd_launch_mixstate_range(*fb->faMix, state0, state1, (hipStream_t)stream);
   if (rc) return rc;
   HIPCHECK(hipEventRecord(fb->ev[5], (hipStream_t)stream));
   return HTKAMD_OK;
}

static int fb_execute_impl(htkamd_fb *fb, const htkamd_fb_config *cfg, htkamd_accs *accs, void *stream, bool defer)
{
   if (!fb || !cfg || !accs) { htkamd_set_error("fb_execute: NULL argument"); return HTKAMD_EINVAL; }
   fb->mixDeferred = false;
   if (accs->m != fb->m) { htkamd_set_error("fb_execute: accumulators belong to a different model"); return HTKAMD_EINVAL; }
   if (fb->nUtt == 0) return HTKAMD_OK;
   const htkamd_model *m = fb->m;
   hipStream_t s = (hipStream_t)stream;
   if (fb->topoVersion != m->topoVersion) {
      htkamd_set_error("fb_execute: the model's minimum durations changed since htkamd_fb_prepare (a re-estimated transition reached or left zero): prepare the batch again");
      return HTKAMD_EINVAL;
   }

   ScoreArgs sa;
   const bool wideTasks = (cfg->scoreMode & (HTKAMD_SCORE_F16 | HTKAMD_SCORE_BF16 | HTKAMD_SCORE_MFMA)) != 0;
   sa.tasks = (const ScoreTask *)(wideTasks ? fb->d_tasksW.p : fb->d_tasks.p); sa.nTasks = (int)(wideTasks ? fb->tasksW.size() : fb->tasks.size()); sa.X = fb->dX;
   sa.slotState = (const int *)fb->d_slotState.p; sa.out = (float *)fb->d_outp.p;
   if (m->NSt > 1) { sa.slotState = (const int *)fb->d_slotStateU.p; sa.out = (float *)fb->d_outpU.p; }      // per (stream, chain state)
   sa.stateCompOff = m->d_stateCompOff; sa.compGauss = m->d_compGauss; sa.compLogWt = m->d_compLogWt;
   sa.gparam = m->d_gparam; sa.PS = m->PS; sa.D = m->D; sa.minLogExp = m->minLogExp;
   sa.laddTab = m->d_laddTab; sa.taskCounter = (int *)fb->d_counter.p;
   if (wideTasks && m->NSt == 1 && fb->wqStart.size() == 9 && !getenv("HTKAMD_NO_XCDQ")) { sa.qStart = (const int *)fb->d_wqStart.p; sa.qCounters = (int *)fb->d_counter.p + 8; }
   if ((cfg->scoreMode & HTKAMD_SCORE_BF16) && !(cfg->scoreMode & HTKAMD_SCORE_F16) && m->NSt == 1 && !getenv("HTKAMD_NO_TAPER_SKIP")) sa.slotRange = (const int *)fb->d_slotRange.p;      // (the switch: for A/B measurements)
   sa.mfmaTab = m->d_mfmaTab; sa.stateTileOff = m->d_stateTileOff; sa.bf16Tab = m->d_bf16Tab; sa.var = m->d_var;
   fb->f16Pass = (cfg->scoreMode & HTKAMD_SCORE_F16) != 0 && !m->tiedMix && sa.nTasks > 0;      // no tasks, no launch: nothing zeroes or raises the flag
   if (fb->f16Pass) {      // the pass's own range flag, behind the status words (zeroed with the task counter before it, by the launcher)
      sa.taskCounter = (int *)((char *)fb->d_pr.p + (sizeof(double) + sizeof(int)) * (size_t)fb->nUtt);
      sa.rangeFlag = sa.taskCounter + 1;
   }
   if (cfg->scoreMode & ~(HTKAMD_SCORE_MFMA | HTKAMD_SCORE_FASTLADD | HTKAMD_SCORE_BF16 | HTKAMD_SCORE_F16 | HTKAMD_SCORE_SOUTP)) { htkamd_set_error("fb_execute: unknown score mode %d", cfg->scoreMode); return HTKAMD_EINVAL; }
   const bool fastLadd = (cfg->scoreMode & HTKAMD_SCORE_FASTLADD) != 0;

   FbArgs fa;
   memset(&fa, 0, sizeof(fa));
   fa.utt = (const UttDesc *)fb->d_utt.p; fa.nUtt = fb->nUtt;
   fa.mN = (const int *)fb->d_mN.p; fa.mTp = (const int *)fb->d_mTp.p; fa.mCell0 = (const int *)fb->d_mCell0.p;
   fa.mSlot0 = (const int *)fb->d_mSlot0.p; fa.mDms = (const int *)fb->d_mDms.p; fa.mHmm = (const int *)fb->d_mHmm.p;
   fa.mTrans = (const int *)fb->d_mTrans.p;
   fa.sQ = (const short *)fb->d_sQ.p;
   fa.thrCell = (const short *)fb->d_thrCell.p; fa.cQ = (const short *)fb->d_cQ.p; fa.cI = (const short *)fb->d_cI.p; fa.slotState = (const int *)fb->d_slotState.p;
   fa.taperLo = (const short *)fb->d_taperLo.p; fa.taperHi = (const short *)fb->d_taperHi.p;
   fa.qLo = (short *)fb->d_qLo.p; fa.qHi = (short *)fb->d_qHi.p; fa.aLo = (short *)fb->d_aLo.p; fa.aHi = (short *)fb->d_aHi.p;
   fa.X = fb->dX; fa.transP = m->d_transP; fa.outp = (float *)fb->d_outp.p;
   fa.beta = (double *)fb->d_beta.p; fa.gam = (double *)fb->d_gam.p; fa.alphaDbg = fb->debug ? (double *)fb->d_alpha.p : nullptr;
   fa.pr = (double *)fb->d_pr.p; fa.status = (int *)((double *)fb->d_pr.p + fb->nUtt);
   fa.stateCompOff = m->d_stateCompOff; fa.compGauss = m->d_compGauss;
   fa.NSt = m->NSt; fa.outpU = (const float *)fb->d_outpU.p; fa.dimStream = m->d_dimStream;
   fa.compatRevisit = compat_revisit(m) && !fb->nextSame.empty(); fa.nextSame = (const int *)fb->d_nextSame.p;
   fa.transOff = (const int *)fb->d_transOff.p; fa.trOccOff = (const int *)fb->d_trOccOff.p;
   fa.compLogWt = m->d_compLogWt; fa.gparam = m->d_gparam; fa.mean = m->d_mean; fa.laddTab = m->d_laddTab;
   fa.PS = m->PS; fa.D = m->D; fa.maxN = m->maxN; fa.maxM = m->maxM;
   fa.nCellsMax = fb->nCellsMax; fa.QMax = fb->QMax; fa.TMax = fb->TMax;
   fa.acc = accs->d_vec; fa.lay = accs->lay;
   fa.pruneInit = cfg->pruneInit; fa.pruneInc = cfg->pruneInc; fa.pruneLim = cfg->pruneLim;
   fa.minLogExp = m->minLogExp; fa.minFrwdP = cfg->minFrwdP; fa.uFlags = cfg->uFlags;
   fa.gamTotal = fb->gamTotal; fa.gamOffByUtt = (const size_t *)fb->d_gamOff.p; fa.gamChunkUtt = (const int *)fb->d_gamChunkUtt.p;

   const size_t nc = fa.nCellsMax, qm = fa.QMax + 3, mn = m->maxN;
   auto r8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
   const size_t ldsTab = r8((size_t)LADD_NK * (LADD_DEG + 1) * 8);
   const size_t ldsBeta = 2 * r8(nc * 8) + r8(qm * 8) + r8(nc * mn * 4) + 4 * r8(qm * 4) + r8(32) + ldsTab;
   const size_t tm = (size_t)fb->TMax + 3;
   const size_t ldsAlpha = 2 * r8(nc * 8) + r8(3 * nc * 8) + r8(qm * 8) + r8(nc * (mn + 1) * 8) + ldsTab + r8(3 * nc * 4) +
                           2 * r8(nc * mn * 4) + 4 * r8(qm * 4) + 2 * r8(tm * 2);

   int rc;
   // the batch tables may have been uploaded on another stream (htkamd_fb_prepare's): the kernels wait for that copy, not the host
   if (fb->copyPending) HIPCHECK(hipStreamWaitEvent(s, fb->evCopy, 0));
   const bool noEv = fb->evMode == 1;      // htkamd_fb_set_event_mode: only the scoring dispatch's own start / stop (no stream events between the kernels)
   HIPCHECK(hipEventRecord(fb->ev[0], s));
   if (m->tiedMix) {
      // hsKind TIEDHS: the pool once per frame, then every state's weighted sum of it (no other arithmetic mode exists for it)
      fa.tmE = (float *)fb->d_tmE.p; fa.tmMaxP = (float *)fb->d_tmMaxP.p; fa.tmPoolOff = m->d_tmPoolOff; fa.tmPool = m->tmPool; fa.totalFrames = fb->totalFrames;
      fa.tmTasks = (const ScoreTask *)fb->d_tasks.p; fa.tmNTasks = (int)fb->tasks.size(); fa.tmSlotState = sa.slotState; fa.tmOut = sa.out;
      fa.compWeight = m->d_compWeight; fa.var = m->d_var;
      if ((rc = htkamd_launch_tm_score(fa, s))) return rc;
      fb->scored = false;
   } else {
   if ((rc = htkamd_launch_score(cfg->scoreMode, m, sa, s, fb->evK[0], fb->evK[1]))) return rc;
   fb->scored = sa.nTasks > 0;
   }
   if (m->NSt > 1) {
      // Setotprob for S > 1 (HFB.c:1057-1066): the state's log probability is the float sum of its streams' in stream order.  The
      // recursions then see a state with "two components" whatever its streams hold (they only ask whether it is a single Gaussian,
      // for the seed they leave to the mixture statistics)
      if ((rc = htkamd_launch_combine_streams(fa, s))) return rc;
      if (m->maxM > 1) fa.stateCompOff = m->d_msCompOff;
   } else if (m->tiedMix) fa.stateCompOff = m->d_msCompOff;
   if (!noEv) HIPCHECK(hipEventRecord(fb->ev[1], s));
   const int nGeneral = fb->clsOff[5] - fb->clsOff[4];
   fb->lastWave = nGeneral == 0;
   if (nGeneral > 0 && ldsAlpha > 160 * 1024) { htkamd_set_error("fb_execute: %zu bytes of LDS needed (max model size %zu states)", ldsAlpha, mn); return HTKAMD_EMODEL; }
   if ((rc = fb->d_betaW.reserve(sizeof(double) * (fb->betaWTotal + 8 * 512)))) return rc;      /* (the lean alpha kernels request whole blocks of eight columns: up to seven columns of 64 W <= 512 values past an utterance's last -- never used, but inside the buffer) */
   fa.betaW = (double *)fb->d_betaW.p;
   const int nLr = fb->clsOff[13] - fb->clsOff[9];
   size_t rows = 0;
   if (nLr > 0) {
      for (int c = 0; c < 4; c++) rows += (size_t)(fb->clsOff[10 + c] - fb->clsOff[9 + c]) * htkamd_stats_lr_chunks(fb->TMax) * (1 << c);
      const size_t nfr = fb->totalFrames ? fb->totalFrames : 1;
      if ((rc = fb->d_laneRec.reserve(sizeof(LaneRec) * (fb->slotState.size() + 1))) || (rc = fb->d_sink.reserve(256))) return rc;
      fa.laneRec = (LaneRec *)fb->d_laneRec.p; fa.sink = (double *)fb->d_sink.p;
      if ((rc = fb->d_alphaW.reserve(sizeof(double) * (fb->alphaWTotal + 64))) || (rc = fb->d_qBeam.reserve(sizeof(int) * (nfr + 2))) ||
          (rc = fb->d_aBeam.reserve(sizeof(int) * (nfr + 2))) || (rc = fb->d_trPart.reserve(sizeof(double) * htkamd_stats_lr_row_doubles() * (rows + 256))) ||
          (rc = fb->d_hits.reserve(sizeof(MixHit) * (rows * htkamd_stats_lr_region_cap() + 64))) || (rc = fb->d_hitCtl.reserve(sizeof(int) * (rows + 1))))
         return rc;
      fa.alphaW = (double *)fb->d_alphaW.p; fa.qBeam = (int *)fb->d_qBeam.p; fa.aBeam = (int *)fb->d_aBeam.p; fa.trPart = (double *)fb->d_trPart.p;
      fa.hits = (MixHit *)fb->d_hits.p; fa.hitCtl = (int *)fb->d_hitCtl.p; fa.nHitRegions = (int)rows; fa.hitRegionCap = htkamd_stats_lr_region_cap();
      fa.hitSlots = (m->NSt > 1 || m->tiedMix) ? 1 : 0;
      // without a pruning beam the beta beams are the taper's, made on the host with the batch tables: the lean beta kernel reads them,
      // and so do the alpha and statistics kernels behind it (fb_lr_lean.inc)
      fa.qBeamNP = (const int *)fb->d_qBeamNP.p;
      if (htkamd_beta_lr_is_lean(fa, fastLadd)) fa.qBeam = (int *)fb->d_qBeamNP.p;
      fb->qBeamLast = fa.qBeam;
      fb->mixStateLast = false;
      { const char *e = getenv("HTKAMD_LR_EXP"); fa.lrExp = e ? atoi(e) : 0; }
      fa.fastMath = fastLadd ? 1 : 0;
      // mixture statistics bucketed by tied state (k_mixstate): the default list mode, one stream, the sparse statistics kernel counting
      static const bool noMixState = getenv("HTKAMD_NO_MIXSTATE") != nullptr;
      if (!noMixState && fb->recCapForce == 0 && !fa.hitSlots && m->maxM <= 16 && (m->D == 39 || m->D == 26 || m->D == 13) && htkamd_stats_lr_is_sparse(fa) &&
          (cfg->uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES))) {
         // room per state: eight times an even share of 2 pairs per frame, a power of two in [64, 65536], the lot within 1 GB
         size_t cap = 64;
         while (cap < 65536 && cap * (size_t)m->S < (size_t)16 * fb->totalFrames) cap <<= 1;
         while (cap > 64 && cap * (size_t)m->S * sizeof(HitS) > ((size_t)1 << 30)) cap >>= 1;
         { const char *e = getenv("HTKAMD_ST_CAP"); if (e && atoi(e) > 0) cap = (size_t)atoi(e); }      // test aid: tiny buckets, so that most pairs take the list kernel behind them
         if ((rc = fb->d_stCnt.reserve(sizeof(int) * (3 * (size_t)m->S + 4))) || (rc = fb->d_stBucket.reserve(sizeof(HitS) * cap * (size_t)m->S))) return rc;
         fb->mixStateLast = true;
         fa.stCnt = (int *)fb->d_stCnt.p; fa.nTiedStates = m->S; fa.stBucket = (HitS *)fb->d_stBucket.p; fa.stCap = (int)cap;
         HIPCHECK(hipMemsetAsync(fa.stCnt, 0, sizeof(int) * (3 * (size_t)m->S + 4), s));  /* counts; pairs turned away (+3 spare); then per state: pairs, triples */
      }
   }
   {  // diagnostic (tools/r06_hosttrace.sh): n more tiny fills in the pass -- what ONE more dispatch costs an iteration on the box at hand
      static const int extra = [] { const char *e = getenv("HTKAMD_EXTRA_FILLS"); return e ? atoi(e) : 0; }();
      for (int i = 0; i < extra; i++) HIPCHECK(hipMemsetAsync(fb->d_counter.p, 0, 4, s));
   }
   static const int clsW[4] = {1, 2, 4, 8};
   // the longest chains first: their recursions are the critical path of the pass
   for (int pass = 0; pass < 2; pass++) {
      FbArgs fc = fa;
      fc.uttList = (const int *)fb->d_uttList.p + fb->clsOff[4]; fc.nList = nGeneral;
      if (nGeneral > 0 && (rc = pass == 0 ? htkamd_launch_beta(fc, fb->blockDim, ldsBeta, s) : htkamd_launch_alpha(fc, fb->blockDim, ldsAlpha, s))) return rc;
      for (int c = 3; c >= 0; c--) {
         fc.uttList = (const int *)fb->d_uttList.p + fb->clsOff[c]; fc.nList = fb->clsOff[c + 1] - fb->clsOff[c];
         if ((rc = pass == 0 ? htkamd_launch_beta_w(fc, clsW[c], fastLadd, s) : htkamd_launch_alpha_w(fc, clsW[c], fastLadd, s))) return rc;
      }
      for (int c = 3; c >= 0; c--) {
         fc.uttList = (const int *)fb->d_uttList.p + fb->clsOff[5 + c]; fc.nList = fb->clsOff[6 + c] - fb->clsOff[5 + c];
         if ((rc = pass == 0 ? htkamd_launch_beta_s(fc, clsW[c], fastLadd, s) : htkamd_launch_alpha_s(fc, clsW[c], fastLadd, s))) return rc;
      }
      for (int c = 3; c >= 0; c--) {
         fc.uttList = (const int *)fb->d_uttList.p + fb->clsOff[9 + c]; fc.nList = fb->clsOff[10 + c] - fb->clsOff[9 + c];
         if ((rc = pass == 0 ? htkamd_launch_beta_lr(fc, clsW[c], fastLadd, s) : htkamd_launch_alpha_lr(fc, clsW[c], fastLadd, s))) return rc;
      }
      if (!noEv) HIPCHECK(hipEventRecord(fb->ev[2 + pass], s));
   }
   {  // left-to-right path: occupation / transition counts and the list of surviving (frame, state) pairs from the stored columns
      FbArgs fc = fa;
      size_t rowOff = 0;
      for (int c = 3; c >= 0; c--) {
         fc.uttList = (const int *)fb->d_uttList.p + fb->clsOff[9 + c]; fc.nList = fb->clsOff[10 + c] - fb->clsOff[9 + c];
         fc.trPart = fa.trPart ? fa.trPart + rowOff * htkamd_stats_lr_row_doubles() : nullptr;
         fc.hits = fa.hits ? fa.hits + rowOff * htkamd_stats_lr_region_cap() : nullptr; fc.hitCtl = fa.hitCtl ? fa.hitCtl + rowOff : nullptr;
         rowOff += (size_t)fc.nList * htkamd_stats_lr_chunks(fb->TMax) * clsW[c];
         if ((rc = htkamd_launch_stats_lr(fc, clsW[c], fastLadd, s))) return rc;
      }
      if (!noEv) HIPCHECK(hipEventRecord(fb->ev[4], s));
   }
   if (cfg->uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS | HTKAMD_UPMIXES)) {
      if ((cfg->uFlags & (HTKAMD_UPMEANS | HTKAMD_UPVARS)) && fb->recCapForce >= 0) {
         // ~4 surviving (frame, component) pairs per frame on a trained system; 16 per frame of room, the rest falls back to atomics
         const size_t cap = fb->recCapForce > 0 ? (size_t)fb->recCapForce : std::min<size_t>((size_t)16 * fb->totalFrames + 4096, (size_t)1 << 30);
         if ((rc = fb->d_rec.reserve(sizeof(MixRec) * cap)) || (rc = fb->d_recSorted.reserve(sizeof(MixRec) * cap)) ||
             (rc = fb->d_recCtl.reserve(sizeof(int) * (3 * ((size_t)m->G + 1) + 1 + (size_t)m->G / 2048 + 2))))
            return rc;
         fa.rec = (MixRec *)fb->d_rec.p; fa.recSorted = (MixRec *)fb->d_recSorted.p; fa.recCap = (int)cap; fa.G = m->G; fa.recCtl = (int *)fb->d_recCtl.p;
      }
      // the dense seed array serves the utterances off the left-to-right path; those on it list their pairs (k_stats_lr -> k_mixhits)
      if (m->tiedMix) {
         fa.stateCompOff = m->d_stateCompOff; fa.rec = nullptr;
         if (fb->nUtt > nLr && (rc = htkamd_launch_mixstats_tm(fa, s))) return rc;
         if (nLr > 0 && (rc = htkamd_launch_mixhits_streams(fa, true, s))) return rc;
      } else if (m->NSt > 1) {
         fa.stateCompOff = m->d_stateCompOff; fa.rec = nullptr;
         if (fb->nUtt > nLr && (rc = htkamd_launch_mixstats_ms(fa, s))) return rc;
         if (nLr > 0 && (rc = htkamd_launch_mixhits_streams(fa, false, s))) return rc;
      }
      else {
         const bool dense = fb->nUtt > nLr && fa.gamTotal > 0, listed = nLr > 0;
         if ((rc = htkamd_launch_mixstats(fa, s, dense, listed, defer))) return rc;
         if (defer && listed && !dense && htkamd_mixstate_applies(fa)) {
            if (!fb->faMix) fb->faMix = new FbArgs();
            *fb->faMix = fa;
            fb->mixDeferred = true;
         }
      }
   }
   HIPCHECK(hipEventRecord(fb->ev[5], s));
   fb->timed = true; fb->evModeLast = noEv ? 1 : 0;
   // the metric's unit count rides along in the accumulator vector so that it is all-reduced with it
   return HTKAMD_OK;
}

static int res_staging(htkamd_fb *fb, size_t bytes)
{
   if (bytes > fb->h_resCap) {
      if (fb->h_res) (void)hipHostFree(fb->h_res);
      fb->h_res = nullptr; fb->h_resCap = 0;
      HIPCHECK(hipHostMalloc(&fb->h_res, bytes + bytes / 4, hipHostMallocDefault));
      fb->h_resCap = bytes + bytes / 4;
   }
   return HTKAMD_OK;
}

// The copy of the results queued on `stream` (the stream of htkamd_fb_execute: it then follows the pass's last kernel in stream order and
// needs no scheduling of its own) and recorded in an event; htkamd_fb_results afterwards only waits for that event.  For host loops that
// queue the next pass before they look at this one's results.
extern "C" int htkamd_fb_results_begin(htkamd_fb *fb, void *stream)
{
   if (!fb) { htkamd_set_error("fb_results_begin: NULL"); return HTKAMD_EINVAL; }
   if (fb->nUtt == 0) return HTKAMD_OK;
   hipStream_t s = (hipStream_t)stream;
   const size_t bytes = (sizeof(double) + sizeof(int)) * (size_t)fb->nUtt + 2 * sizeof(int);
   int rc = res_staging(fb, bytes);
   if (rc) return rc;
   if (!fb->evRes) HIPCHECK(hipEventCreateWithFlags(&fb->evRes, hipEventDisableTiming));
   HIPCHECK(hipMemcpyAsync(fb->h_res, fb->d_pr.p, bytes, hipMemcpyDeviceToHost, s));
   HIPCHECK(hipEventRecord(fb->evRes, s));
   fb->resPending = true;
   return HTKAMD_OK;
}

extern "C" int htkamd_fb_results(htkamd_fb *fb, double *pr, int *status, void *stream)
{
   if (!fb) { htkamd_set_error("fb_results: NULL"); return HTKAMD_EINVAL; }
   hipStream_t s = (hipStream_t)stream;
   if (fb->nUtt == 0) return HTKAMD_OK;
   // log probabilities and status words lie in one device buffer and come back in one copy through a pinned staging buffer
   const size_t bytes = (sizeof(double) + sizeof(int)) * (size_t)fb->nUtt + 2 * sizeof(int);
   if (fb->resPending) {                                    // htkamd_fb_results_begin queued the copy behind the pass
      fb->resPending = false;
      HIPCHECK(hipEventSynchronize(fb->evRes));
   } else {
      int rc = res_staging(fb, bytes);
      if (rc) return rc;
      if (fb->timed) {
         // wait for THIS batch's last kernel only (work queued on the stream afterwards, e.g. the next batch, keeps running)
         HIPCHECK(hipEventSynchronize(fb->ev[5]));
         HIPCHECK(hipMemcpyAsync(fb->h_res, fb->d_pr.p, bytes, hipMemcpyDeviceToHost, fb->resStream));
         HIPCHECK(hipStreamSynchronize(fb->resStream));
      } else {
         HIPCHECK(hipMemcpyAsync(fb->h_res, fb->d_pr.p, bytes, hipMemcpyDeviceToHost, s));
         HIPCHECK(hipStreamSynchronize(s));
      }
   }
   if (pr) memcpy(pr, fb->h_res, sizeof(double) * (size_t)fb->nUtt);
   if (status) memcpy(status, (const char *)fb->h_res + sizeof(double) * (size_t)fb->nUtt, sizeof(int) * (size_t)fb->nUtt);
   if (fb->f16Pass) {
      int flag;
      memcpy(&flag, (const char *)fb->h_res + (sizeof(double) + sizeof(int)) * (size_t)fb->nUtt + sizeof(int), sizeof(int));
      if (flag) {
         htkamd_set_error("fb_results: the fp16 scoring path met %s%s%s outside its range: nothing of this pass can be used (accumulators included) -- repeat it with HTKAMD_SCORE_BF16",
                          (flag & HTKAMD_F16_EMODEL) ? "a model coefficient" : "", (flag & HTKAMD_F16_EMODEL) && (flag & HTKAMD_F16_EFEAT) ? " and " : "", (flag & HTKAMD_F16_EFEAT) ? "a feature value" : "");
         return HTKAMD_ERANGE;
      }
   }
   return HTKAMD_OK;
}

// The events htkamd_fb_execute records for htkamd_fb_kernel_times*: 0 (default) the scoring dispatch's own start / stop and five stream
// events between the kernels (each a barrier packet: ~5 us of idle queue apiece, 20-40 us of a 2 ms pass); 1 the scoring dispatch's only --
// htkamd_fb_kernel_times5 then reports -1 for the other four intervals.
extern "C" int htkamd_fb_set_event_mode(htkamd_fb *fb, int mode)
{
   if (!fb || mode < 0 || mode > 1) { htkamd_set_error("fb_set_event_mode: bad argument"); return HTKAMD_EINVAL; }
   fb->evMode = mode;
   return HTKAMD_OK;
}

// out[0..4]: scoring (the dispatch's own start -> stop), beta, alpha, the left-to-right path's frame-parallel statistics, mixture
// statistics -- the last four as intervals between stream events around the launches
extern "C" int htkamd_fb_kernel_times5(htkamd_fb *fb, double out[5])
{
   if (!fb || !out) { htkamd_set_error("fb_kernel_times: NULL"); return HTKAMD_EINVAL; }
   if (!fb->timed) { htkamd_set_error("fb_kernel_times: nothing executed yet"); return HTKAMD_EINVAL; }
   HIPCHECK(hipEventSynchronize(fb->ev[5]));
   for (int i = 0; i < 5; i++) {
      float ms = 0.f;
      if (i == 0 && fb->scored) HIPCHECK(hipEventElapsedTime(&ms, fb->evK[0], fb->evK[1]));
      else if (fb->evModeLast == 1) { out[i] = -1.0; continue; }                  // not measured in this pass
      else HIPCHECK(hipEventElapsedTime(&ms, fb->ev[i], fb->ev[i + 1]));
      out[i] = (double)ms * 1e-3;
   }
   return HTKAMD_OK;
}

// out[0]: (frame, state) pairs the MINFORPROB prune let through in the last pass, out[1]: (frame, state, component) triples whose
// statistics were accumulated -- the units of the mixture-statistics kernel's roofline (2 D FMAs and 624 bytes of
// accumulator traffic per triple at D = 39).  Counted by k_mixstate; -1 where the pass ran on the list kernels alone.
extern "C" int htkamd_fb_mix_counts(htkamd_fb *fb, long long out[2])
{
   if (!fb || !out) { htkamd_set_error("fb_mix_counts: NULL"); return HTKAMD_EINVAL; }
   out[0] = out[1] = -1;
   if (!fb->timed || !fb->mixStateLast || !fb->d_stCnt.p) return HTKAMD_OK;
   HIPCHECK(hipEventSynchronize(fb->ev[5]));
   const size_t S = (size_t)fb->m->S;
   int *c = (int *)malloc(sizeof(int) * 2 * S);
   if (!c) { htkamd_set_error("fb_mix_counts: out of memory"); return HTKAMD_ENOMEM; }
   const hipError_t e = hipMemcpy(c, (int *)fb->d_stCnt.p + S + 4, sizeof(int) * 2 * S, hipMemcpyDeviceToHost);
   if (e != hipSuccess) { free(c); HIPCHECK(e); }
   out[0] = out[1] = 0;
   for (size_t i = 0; i < S; i++) { out[0] += c[2 * i]; out[1] += c[2 * i + 1]; }
   free(c);
   return HTKAMD_OK;
}

// the four intervals of round 1/2: scoring, beta, alpha + occupation statistics, mixture statistics
extern "C" int htkamd_fb_kernel_times(htkamd_fb *fb, double out[4])
{
   double t[5];
   const int rc = htkamd_fb_kernel_times5(fb, t);
   if (rc) return rc;
   // (-1: an interval the pass did not measure, htkamd_fb_set_event_mode(fb, 1) -- it stays -1 here, a sum with one is -1 too)
   out[0] = t[0]; out[1] = t[1]; out[2] = (t[2] < 0.0 || t[3] < 0.0) ? -1.0 : t[2] + t[3]; out[3] = t[4];
   return HTKAMD_OK;
}

extern "C" int htkamd_fb_get_trellis(htkamd_fb *fb, int u, double *beta, double *alpha, float *outp,
                                     int *qLo, int *qHi, int *aLo, int *aHi, int *pT, int *pQ, int *pMaxN, void *stream)
{
   if (!fb || u < 0 || u >= fb->nUtt) { htkamd_set_error("fb_get_trellis: bad utterance index"); return HTKAMD_EINVAL; }
   hipStream_t s = (hipStream_t)stream;
   HIPCHECK(hipStreamSynchronize(s));
   const UttDesc &d = fb->utt[u];
   const int T = d.T, Q = d.Q, nC = d.nCells, nS = d.nSlots, maxN = fb->m->maxN;
   if (pT) *pT = T; if (pQ) *pQ = Q; if (pMaxN) *pMaxN = maxN;
   if (T <= 0 || Q <= 0) return HTKAMD_OK;
   std::vector<short> lo(T), hi(T), alo(T), ahi(T);
   HIPCHECK(hipMemcpy(lo.data(), (short *)fb->d_qLo.p + d.frame0, sizeof(short) * T, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(hi.data(), (short *)fb->d_qHi.p + d.frame0, sizeof(short) * T, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(alo.data(), (short *)fb->d_aLo.p + d.frame0, sizeof(short) * T, hipMemcpyDeviceToHost));
   HIPCHECK(hipMemcpy(ahi.data(), (short *)fb->d_aHi.p + d.frame0, sizeof(short) * T, hipMemcpyDeviceToHost));
   if (d.W > 0 && d.pad == 2 && fb->qBeamLast) {         // left-to-right path: the beta beams as words (the lean beta kernel writes no qLo / qHi)
      std::vector<int> qb(T);
      HIPCHECK(hipMemcpy(qb.data(), fb->qBeamLast + d.frame0, sizeof(int) * T, hipMemcpyDeviceToHost));
      for (int t = 0; t < T; t++) { lo[t] = (short)(qb[t] & 0xffff); hi[t] = (short)((qb[t] >> 16) & 0xffff); }
   }
   if (d.W > 0 && d.pad == 2) {                          // left-to-right path: the alpha beam comes as lanes of its first and last model
      std::vector<int> ab(T);
      HIPCHECK(hipMemcpy(ab.data(), (int *)fb->d_aBeam.p + d.frame0, sizeof(int) * T, hipMemcpyDeviceToHost));
      for (int t = 0; t < T; t++) { alo[t] = fb->sQ[d.slot0 + (ab[t] & 0xffff)]; ahi[t] = fb->sQ[d.slot0 + ((ab[t] >> 16) & 0xffff)]; }
   }
   for (int t = 0; t < T; t++) {
      if (qLo) qLo[t] = lo[t]; if (qHi) qHi[t] = hi[t]; if (aLo) aLo[t] = alo[t]; if (aHi) aHi[t] = ahi[t];
   }
   const int *mN = fb->mN.data() + d.q0, *mC = fb->mCell0.data() + d.q0, *mS = fb->mSlot0.data() + d.q0;
   const size_t n = (size_t)T * Q * maxN;
   if (beta) {
      std::vector<double> b((size_t)T * nC);
      if (d.W > 0 && d.pad == 2) {                       // left-to-right path: betaS[T][L] only; the entry state's value is a_12 + b_2 + beta_2
         const size_t Lw = (size_t)64 * d.W;             // (SetBeta with one entry transition), the exit state's the next model's entry value one frame on
         std::vector<double> bs((size_t)T * Lw);
         HIPCHECK(hipMemcpy(bs.data(), (double *)fb->d_betaW.p + d.betaW0, sizeof(double) * bs.size(), hipMemcpyDeviceToHost));
         std::vector<float> o((size_t)T * nS);
         HIPCHECK(hipMemcpy(o.data(), (float *)fb->d_outp.p + d.outp0, sizeof(float) * o.size(), hipMemcpyDeviceToHost));
         const int *mSl = fb->mSlot0.data() + d.q0, *mTp = fb->mTp.data() + d.q0;
         auto entry = [&](int t, int q) {                // beta_1(q, t) (t 0-based)
            const int l0 = mSl[q - 1];
            const double aa = fb->m->h_transP[mTp[q - 1] + 1], y = bs[(size_t)t * Lw + l0];
            if (!(aa > LSMALL && y > LSMALL)) return (double)LZERO;
            const double v = aa + (double)o[(size_t)l0 * T + t] + y;
            return v < LSMALL ? (double)LZERO : v;
         };
         for (int t = 0; t < T; t++)
            for (int q = 1; q <= Q; q++) {
               const int Nq = mN[q - 1], l0 = mSl[q - 1];
               double *cell = &b[(size_t)t * nC + mC[q - 1]];
               cell[0] = (q >= lo[t] && q <= hi[t]) ? entry(t, q) : LZERO;
               for (int i = 2; i < Nq; i++) cell[i - 1] = bs[(size_t)t * Lw + l0 + i - 2];
               double bN = LZERO;
               if (t == T - 1) bN = (q == Q) ? 0.0 : LZERO;
               else if (q < Q && q + 1 >= lo[t + 1] && q + 1 <= hi[t + 1]) bN = entry(t + 1, q + 1);
               cell[Nq - 1] = bN;
            }
      } else if (d.W > 0 && d.pad == 1) {                // state-per-lane path: betaS[T][L] (emitting states) + betaE[T][L] (entry state at the
         const size_t Lw = (size_t)64 * d.W;             // model's first lane); the exit state's value is the next model's entry value one frame on
         std::vector<double> bs((size_t)T * 2 * Lw);
         HIPCHECK(hipMemcpy(bs.data(), (double *)fb->d_betaW.p + d.betaW0, sizeof(double) * bs.size(), hipMemcpyDeviceToHost));
         const int *mSl = fb->mSlot0.data() + d.q0;
         for (int t = 0; t < T; t++)
            for (int q = 1; q <= Q; q++) {
               const int Nq = mN[q - 1], l0 = mSl[q - 1];
               double *cell = &b[(size_t)t * nC + mC[q - 1]];
               cell[0] = bs[(size_t)T * Lw + (size_t)t * Lw + l0];
               for (int i = 2; i < Nq; i++) cell[i - 1] = bs[(size_t)t * Lw + l0 + i - 2];
               double bN = LZERO;
               if (t == T - 1) bN = (q == Q) ? 0.0 : LZERO;
               else if (q < Q && q + 1 >= lo[t + 1] && q + 1 <= hi[t + 1]) bN = bs[(size_t)T * Lw + (size_t)(t + 1) * Lw + mSl[q]];
               cell[Nq - 1] = bN;
            }
      } else if (d.W > 0) {                              // the wave path's block [frame][state][lane] -> cells
         const size_t run = (size_t)64 * d.W;
         std::vector<double> bs((size_t)T * 5 * run);
         HIPCHECK(hipMemcpy(bs.data(), (double *)fb->d_betaW.p + d.betaW0, sizeof(double) * bs.size(), hipMemcpyDeviceToHost));
         for (int t = 0; t < T; t++)
            for (int q = 1; q <= Q; q++)
               for (int i = 0; i < mN[q - 1]; i++)
                  b[(size_t)t * nC + mC[q - 1] + i] = bs[((size_t)t * 5 + i) * run + (q - 1)];
      } else
      HIPCHECK(hipMemcpy(b.data(), (double *)fb->d_beta.p + d.beta0, sizeof(double) * b.size(), hipMemcpyDeviceToHost));
      for (size_t k = 0; k < n; k++) beta[k] = NAN;
      for (int t = 0; t < T; t++)
         for (int q = lo[t]; q <= hi[t]; q++)
            for (int i = 0; i < mN[q - 1]; i++)
               beta[((size_t)t * Q + (q - 1)) * maxN + i] = b[(size_t)t * nC + mC[q - 1] + i];
   }
   if (alpha) {
      if (!fb->debug) { htkamd_set_error("fb_get_trellis: alpha is only kept in debug mode (htkamd_fb_set_debug before prepare)"); return HTKAMD_EINVAL; }
      std::vector<double> al((size_t)T * nC);
      HIPCHECK(hipMemcpy(al.data(), (double *)fb->d_alpha.p + d.beta0, sizeof(double) * al.size(), hipMemcpyDeviceToHost));
      for (size_t k = 0; k < n; k++) alpha[k] = NAN;
      for (int t = 0; t < T; t++)
         for (int q = 1; q <= Q; q++)
            for (int i = 0; i < mN[q - 1]; i++)
               alpha[((size_t)t * Q + (q - 1)) * maxN + i] = al[(size_t)t * nC + mC[q - 1] + i];
   }
   if (outp) {
      std::vector<float> o((size_t)T * nS);
      HIPCHECK(hipMemcpy(o.data(), (float *)fb->d_outp.p + d.outp0, sizeof(float) * o.size(), hipMemcpyDeviceToHost));
      for (size_t k = 0; k < n; k++) outp[k] = NAN;
      for (int t = 0; t < T; t++)
         for (int q = 1; q <= Q; q++)
            for (int j = 2; j < mN[q - 1]; j++)
               outp[((size_t)t * Q + (q - 1)) * maxN + (j - 1)] = o[(size_t)(mS[q - 1] + j - 2) * T + t];
   }
   return HTKAMD_OK;
}
