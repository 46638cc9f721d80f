// comm.hip -- the one exchange step of a multi-GPU HERest pass, callable from C: sum all-reduce of the accumulator vector over RCCL.
//
// Replaces HERest's parallel mode round trip -- every process `HERest -p k` dumps HERk.acc (DumpAccs HTrain.c:1453), one process
// `HERest -p 0` loads and adds them all (LoadAccs HTrain.c:1625, HERest.c:514-550) -- with ncclAllReduce(sum) on the flat fp64 vector
// where it lies in HBM: one process per GPU, intra-node xGMI.  After it every rank holds the same sums and applies the same update
// (htkamd_model_update_device), so no model has to be sent anywhere.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a host that brings its own RCCL (PyTorch ships one) gets that copy, a plain C
// host the one of the ROCm installation, and a single-GPU host needs none.  Rendezvous is the application's: rank 0 obtains a
// 128-byte id (htkamd_comm_unique_id) and hands it to the other ranks by whatever channel it has (tools/herest.c: a file).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "internal.h"
#include "hipcheck.h"

typedef struct { char internal[128]; } rcclId;                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void *rcclComm;
typedef int (*fnGetUniqueId)(rcclId *);
typedef int (*fnCommInitRank)(rcclComm *, int, rcclId, int);
typedef int (*fnAllReduce)(const void *, void *, size_t, int, int, rcclComm, hipStream_t);
typedef int (*fnCommDestroy)(rcclComm);
typedef const char *(*fnGetErrorString)(int);

static struct { void *h; fnGetUniqueId getId; fnCommInitRank init; fnAllReduce allReduce; fnCommDestroy destroy; fnGetErrorString err; } g_rccl;

static int rccl_bind()
{
   if (g_rccl.h) return HTKAMD_OK;
   const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
   void *h = nullptr;
   for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
   if (!h) { htkamd_set_error("comm: RCCL not found (librccl.so.1): %s", dlerror()); return HTKAMD_ENODEV; }
   g_rccl.getId = (fnGetUniqueId)dlsym(h, "ncclGetUniqueId"); g_rccl.init = (fnCommInitRank)dlsym(h, "ncclCommInitRank");
   g_rccl.allReduce = (fnAllReduce)dlsym(h, "ncclAllReduce"); g_rccl.destroy = (fnCommDestroy)dlsym(h, "ncclCommDestroy");
   g_rccl.err = (fnGetErrorString)dlsym(h, "ncclGetErrorString");
   if (!g_rccl.getId || !g_rccl.init || !g_rccl.allReduce || !g_rccl.destroy) { htkamd_set_error("comm: librccl lacks the expected entry points"); dlclose(h); return HTKAMD_ENODEV; }
   g_rccl.h = h;
   return HTKAMD_OK;
}

struct htkamd_comm { rcclComm c; int nRanks, rank; float *d_wire; size_t wireCap, wireUsed; int *d_flag; };

#define RCCLCHECK(call) do { const int r_ = (call); if (r_ != 0) { htkamd_set_error("%s -> %s", #call, g_rccl.err ? g_rccl.err(r_) : "RCCL error"); return HTKAMD_EHIP; } } while (0)

extern "C" int htkamd_comm_unique_id(void *id128)
{
   if (!id128) { htkamd_set_error("comm_unique_id: NULL"); return HTKAMD_EINVAL; }
   int rc = rccl_bind();
   if (rc) return rc;
   rcclId id;
   RCCLCHECK(g_rccl.getId(&id));
   memcpy(id128, &id, sizeof(id));
   return HTKAMD_OK;
}

extern "C" int htkamd_comm_init(htkamd_comm **out, int nRanks, int rank, const void *id128)
{
   if (!out || nRanks < 1 || rank < 0 || rank >= nRanks || (nRanks > 1 && !id128)) { htkamd_set_error("comm_init: bad argument"); return HTKAMD_EINVAL; }
   htkamd_comm *c = (htkamd_comm *)calloc(1, sizeof(htkamd_comm));
   c->nRanks = nRanks; c->rank = rank;
   if (nRanks > 1) {
      int rc = rccl_bind();
      if (rc) { free(c); return rc; }
      rcclId id;
      memcpy(&id, id128, sizeof(id));
      const int r = g_rccl.init(&c->c, nRanks, id, rank);
      if (r != 0) { htkamd_set_error("comm_init: ncclCommInitRank -> %s", g_rccl.err ? g_rccl.err(r) : "RCCL error"); free(c); return HTKAMD_EHIP; }
   }
   *out = c;
   return HTKAMD_OK;
}

extern "C" void htkamd_comm_destroy(htkamd_comm *c)
{
   if (!c) return;
   if (c->c && g_rccl.destroy) (void)g_rccl.destroy(c->c);
   if (c->d_wire) (void)hipFree(c->d_wire);
   if (c->d_flag) (void)hipFree(c->d_flag);
   free(c);
}

extern "C" int htkamd_comm_ranks(const htkamd_comm *c) { return c ? c->nRanks : 1; }

// In-place sum over the ranks of the whole vector, asynchronous on `stream`.  One rank: nothing to do.
extern "C" int htkamd_accs_allreduce(htkamd_accs *a, htkamd_comm *c, void *stream)
{
   if (!a || !c) { htkamd_set_error("accs_allreduce: NULL argument"); return HTKAMD_EINVAL; }
   if (c->nRanks == 1) return HTKAMD_OK;
   RCCLCHECK(g_rccl.allReduce(a->d_vec, a->d_vec, a->lay.total, 8 /* ncclFloat64 */, 0 /* ncclSum */, c->c, (hipStream_t)stream));
   return HTKAMD_OK;
}

// The same exchange with the statistics on the wire as fp32 (HTKAMD_WIRE_F32): every rank rounds its fp64 partial sums to float ONCE,
// the ring adds floats, the result goes back into the fp64 vector.  Half the bytes of the fp64 exchange (SURVEY §8(e) prices the
// payload as 25.9 MB of floats); still tighter than the reference, whose accumulators ARE floats summed utterance by utterance
// (HTrain.c:1625-1687 adds float dumps).  The counters behind the statistics -- nEgs, totalPr, totalT, utterance and evaluation
// counts: integers beyond 2^24 and a sum of ~1e8 -- stay fp64 in a second, small all-reduce.
__global__ void k_wire_pack(const double *__restrict__ v, float *__restrict__ w, size_t n)
{
   for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = (float)v[i];
}
__global__ void k_wire_unpack(const float *__restrict__ w, double *__restrict__ v, size_t n)
{
   for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v[i] = (double)w[i];
}

extern "C" int htkamd_accs_wire_round(htkamd_accs *a, void *stream)
{
   // what HTKAMD_WIRE_F32 does to ONE rank's vector before the sum (for tests that emulate the exchange without RCCL)
   if (!a) { htkamd_set_error("accs_wire_round: NULL argument"); return HTKAMD_EINVAL; }
   const size_t bulk = a->lay.nEgs;
   float *w = nullptr;
   HIPCHECK(hipMalloc(&w, sizeof(float) * bulk));
   k_wire_pack<<<1024, 256, 0, (hipStream_t)stream>>>(a->d_vec, w, bulk);
   HIPCHECK(hipGetLastError());
   k_wire_unpack<<<1024, 256, 0, (hipStream_t)stream>>>(w, a->d_vec, bulk);
   HIPCHECK(hipGetLastError());
   HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
   HIPCHECK(hipFree(w));
   return HTKAMD_OK;
}

extern "C" int htkamd_accs_allreduce_wire(htkamd_accs *a, htkamd_comm *c, int wire, void *stream)
{
   if (wire == HTKAMD_WIRE_F64) return htkamd_accs_allreduce(a, c, stream);
   if (!a || !c || wire != HTKAMD_WIRE_F32) { htkamd_set_error("accs_allreduce_wire: bad argument"); return HTKAMD_EINVAL; }
   if (c->nRanks == 1) return HTKAMD_OK;
   const size_t bulk = a->lay.nEgs, tail = a->lay.total - bulk;
   if (c->wireCap < bulk) {
      if (c->d_wire) HIPCHECK(hipFree(c->d_wire));
      c->d_wire = nullptr; c->wireCap = 0;
      HIPCHECK(hipMalloc(&c->d_wire, sizeof(float) * bulk));
      c->wireCap = bulk;
   }
   hipStream_t st = (hipStream_t)stream;
   k_wire_pack<<<1024, 256, 0, st>>>(a->d_vec, c->d_wire, bulk);
   HIPCHECK(hipGetLastError());                          // (a pack that did not launch would let RCCL sum an uninitialised buffer into the statistics)
   RCCLCHECK(g_rccl.allReduce(c->d_wire, c->d_wire, bulk, 7 /* ncclFloat32 */, 0 /* ncclSum */, c->c, st));
   RCCLCHECK(g_rccl.allReduce(a->d_vec + bulk, a->d_vec + bulk, tail, 8 /* ncclFloat64 */, 0, c->c, st));
   k_wire_unpack<<<1024, 256, 0, st>>>(c->d_wire, a->d_vec, bulk);
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

// ---- the exchange in parts (htkamd_fb_execute_begin / _mix): ranges of the vector by tied state
extern "C" int htkamd_accs_state_ranges(htkamd_accs *a, int state0, int state1, int withRest, size_t off[7], size_t len[7], int *n)
{
   if (!a || !off || !len || !n) { htkamd_set_error("accs_state_ranges: NULL argument"); return HTKAMD_EINVAL; }
   const htkamd_model *m = a->m;
   if (state0 < 0 || state1 > m->S || state0 > state1) { htkamd_set_error("accs_state_ranges: states [%d, %d) outside the set's %d", state0, state1, m->S); return HTKAMD_EINVAL; }
   if (a->stateOrder == 0) {
      a->stateOrder = (m->G == m->C && m->NSt <= 1 && !m->tiedMix) ? 1 : -1;
      for (int c = 0; c < m->C && a->stateOrder == 1; c++) if (m->h_compGauss[c] != c) a->stateOrder = -1;
   }
   if (a->stateOrder < 0) { htkamd_set_error("accs_state_ranges: the set's components do not own their Gaussians in state order (shared pdfs, several streams or tied mixtures): exchange the vector whole"); return HTKAMD_EMODEL; }
   const size_t g0 = (size_t)m->h_stateCompOff[state0], g1 = (size_t)m->h_stateCompOff[state1], D = (size_t)m->D;
   int k = 0;
   auto put = [&](size_t o, size_t l) { if (l) { off[k] = o; len[k] = l; k++; } };
   put(a->lay.mu + g0 * D, (g1 - g0) * D); put(a->lay.muOcc + g0, g1 - g0);
   put(a->lay.va + g0 * D, (g1 - g0) * D); put(a->lay.vaOcc + g0, g1 - g0);
   put(a->lay.wt + g0, g1 - g0); put(a->lay.wtOcc + (size_t)state0, (size_t)(state1 - state0));
   if (withRest) put(a->lay.tr, a->lay.nEgs - a->lay.tr);                      // tr, trOcc: up to the counters
   *n = k;
   return HTKAMD_OK;
}

struct RangeSet { size_t off[7], len[7], start[8]; int n; };
template <typename W, bool PACK>
__global__ void k_ranges(double *__restrict__ v, W *__restrict__ w, RangeSet r)
{
   const size_t total = r.start[r.n];
   for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      int k = 0;
      while (k + 1 < r.n && i >= r.start[k + 1]) k++;
      const size_t j = r.off[k] + (i - r.start[k]);
      if (PACK) w[i] = (W)v[j]; else v[j] = (double)w[i];
   }
}

static int ranges_launch(htkamd_accs *a, int n, const size_t *off, const size_t *len, int wire, void *buf, hipStream_t st, bool pack, size_t *totalOut)
{
   if (!a || n < 0 || n > 7 || (n && (!off || !len)) || (wire != HTKAMD_WIRE_F32 && wire != HTKAMD_WIRE_F64)) { htkamd_set_error("accs ranges: bad argument"); return HTKAMD_EINVAL; }
   RangeSet r; r.n = n; r.start[0] = 0;
   for (int k = 0; k < n; k++) {
      if (off[k] + len[k] > a->lay.total) { htkamd_set_error("accs ranges: range %d beyond the vector", k); return HTKAMD_EINVAL; }
      r.off[k] = off[k]; r.len[k] = len[k]; r.start[k + 1] = r.start[k] + len[k];
   }
   if (totalOut) *totalOut = r.start[n];
   if (r.start[n] == 0) return HTKAMD_OK;
   if (!buf) { htkamd_set_error("accs ranges: NULL buffer"); return HTKAMD_EINVAL; }
   const unsigned blocks = (unsigned)std::min<size_t>((r.start[n] + 255) / 256, 2048);
   if (wire == HTKAMD_WIRE_F32) { if (pack) k_ranges<float, true><<<blocks, 256, 0, st>>>(a->d_vec, (float *)buf, r); else k_ranges<float, false><<<blocks, 256, 0, st>>>(a->d_vec, (float *)buf, r); }
   else { if (pack) k_ranges<double, true><<<blocks, 256, 0, st>>>(a->d_vec, (double *)buf, r); else k_ranges<double, false><<<blocks, 256, 0, st>>>(a->d_vec, (double *)buf, r); }
   HIPCHECK(hipGetLastError());
   return HTKAMD_OK;
}

extern "C" int htkamd_accs_pack_ranges(htkamd_accs *a, int n, const size_t *off, const size_t *len, int wire, void *dst, void *stream)
{
   return ranges_launch(a, n, off, len, wire, dst, (hipStream_t)stream, true, nullptr);
}

extern "C" int htkamd_accs_unpack_ranges(htkamd_accs *a, int n, const size_t *off, const size_t *len, int wire, const void *src, void *stream)
{
   return ranges_launch(a, n, off, len, wire, (void *)src, (hipStream_t)stream, false, nullptr);
}

extern "C" int htkamd_accs_allreduce_states(htkamd_accs *a, htkamd_comm *c, int wire, int state0, int state1, int withRest, void *stream)
{
   if (!a || !c) { htkamd_set_error("accs_allreduce_states: NULL argument"); return HTKAMD_EINVAL; }
   if (c->nRanks == 1) return HTKAMD_OK;
   size_t off[7], len[7], total = 0;
   int n = 0;
   int rc = htkamd_accs_state_ranges(a, state0, state1, withRest, off, len, &n);
   if (rc) return rc;
   for (int k = 0; k < n; k++) total += len[k];
   hipStream_t st = (hipStream_t)stream;
   // the part's room in the wire buffer: parts of one iteration lie one behind the other (a part may still be on its way when the next is packed
   // on another stream); a call with withRest closes the iteration
   const size_t need = a->lay.nEgs * (wire == HTKAMD_WIRE_F64 ? 2 : 1);             // in floats
   if (c->wireCap < need) {
      if (c->d_wire) HIPCHECK(hipFree(c->d_wire));
      c->d_wire = nullptr; c->wireCap = 0; c->wireUsed = 0;
      HIPCHECK(hipMalloc(&c->d_wire, sizeof(float) * need));
      c->wireCap = need;
   }
   const size_t words = total * (wire == HTKAMD_WIRE_F64 ? 2 : 1);
   if (c->wireUsed + words > c->wireCap) c->wireUsed = 0;
   void *buf = c->d_wire + c->wireUsed;
   c->wireUsed = withRest ? 0 : c->wireUsed + ((words + 63) & ~(size_t)63);
   if (total) {
      if ((rc = ranges_launch(a, n, off, len, wire, buf, st, true, nullptr))) return rc;
      RCCLCHECK(g_rccl.allReduce(buf, buf, total, wire == HTKAMD_WIRE_F64 ? 8 : 7, 0 /* ncclSum */, c->c, st));
      if ((rc = ranges_launch(a, n, off, len, wire, buf, st, false, nullptr))) return rc;
   }
   if (withRest) RCCLCHECK(g_rccl.allReduce(a->d_vec + a->lay.nEgs, a->d_vec + a->lay.nEgs, a->lay.total - a->lay.nEgs, 8 /* ncclFloat64 */, 0, c->c, st));
   return HTKAMD_OK;
}

// max over the ranks of one int, synchronous: the ranks agree on a decision every one of them must take alike (tools/herest.c: the
// fp16 -> bf16 fallback of an iteration, so that no two ranks add statistics of different arithmetic)
extern "C" int htkamd_comm_agree_max(htkamd_comm *c, int *value, void *stream)
{
   if (!c || !value) { htkamd_set_error("comm_agree_max: NULL argument"); return HTKAMD_EINVAL; }
   if (c->nRanks == 1) return HTKAMD_OK;
   if (!c->d_flag) HIPCHECK(hipMalloc(&c->d_flag, sizeof(int)));
   hipStream_t st = (hipStream_t)stream;
   HIPCHECK(hipMemcpyAsync(c->d_flag, value, sizeof(int), hipMemcpyHostToDevice, st));
   RCCLCHECK(g_rccl.allReduce(c->d_flag, c->d_flag, 1, 2 /* ncclInt32 */, 2 /* ncclMax */, c->c, st));
   HIPCHECK(hipMemcpyAsync(value, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPCHECK(hipStreamSynchronize(st));
   return HTKAMD_OK;
}
