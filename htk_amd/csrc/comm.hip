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

struct htkamd_comm { rcclComm c; int nRanks, rank; };

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
