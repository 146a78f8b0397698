// Gradient exchange over RCCL bound DIRECTLY (no torch.distributed in between): one communicator per process, the flat gradient
// buffers SUM-all-reduced in place on a caller-given HIP stream -- the replacement of nn.DataParallel's gather / scatter per call
// (reference src/train.py:269-274; SURVEY.md 8(e): one gradient all-reduce per iteration, batch-sharded replicas).
//
// Why not torch.distributed's ProcessGroupNCCL: its watchdog thread polls HIP events, and an event query that lands while another
// thread captures a stream kills the capture (hipErrorStreamCaptureUnsupported), so the collectives could not live INSIDE the
// captured training iteration and rsis_amd.train.GraphedStep had to cut the iteration into 3-4 graphs with the collectives launched
// eagerly between them (~0.3-0.45 ms per cut).  A collective issued through these entry points is an ordinary stream operation:
// it is captured into the iteration's hipGraph like any kernel (on a forked stream, so that it overlaps the rest of the backward).
// RCCL is resolved at run time (dlopen of the librccl the process already has -- torch ships one -- else /opt/rocm's): the library
// has no link-time dependency on it and single-GPU users never load it.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/rsis_hip.h"

namespace {
typedef struct { char internal[128]; } nccl_uid;     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h)
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_destroy)(nccl_comm);
typedef int (*fn_count)(nccl_comm, int*);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_count count = nullptr;
  fn_errstr errstr = nullptr;
  bool ok = false;
};
Rccl g_rccl;
char g_last_error[256] = "";

bool load_rccl() {
  if (g_rccl.ok) return true;
  if (g_rccl.handle == nullptr) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !g_rccl.handle; ++pass)           // first: a copy the process has already loaded (torch's)
      for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (g_rccl.handle) break;
      }
    if (!g_rccl.handle) { snprintf(g_last_error, sizeof g_last_error, "librccl not found: %s", dlerror()); return false; }
  }
  g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.destroy = (fn_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.count = (fn_count)dlsym(g_rccl.handle, "ncclCommCount");
  g_rccl.errstr = (fn_errstr)dlsym(g_rccl.handle, "ncclGetErrorString");
  g_rccl.ok = g_rccl.get_uid && g_rccl.init_rank && g_rccl.allreduce && g_rccl.destroy && g_rccl.count;
  if (!g_rccl.ok) snprintf(g_last_error, sizeof g_last_error, "librccl lacks a required symbol");
  return g_rccl.ok;
}
int fail(const char* what, int rc) {
  snprintf(g_last_error, sizeof g_last_error, "%s: %s (ncclResult %d)", what, g_rccl.errstr ? g_rccl.errstr(rc) : "?", rc);
  return RSIS_ERR_LAUNCH;
}
}  // namespace

extern "C" {
const char* rsis_comm_last_error(void) { return g_last_error; }

int rsis_comm_available(void) { return load_rccl() ? RSIS_OK : RSIS_ERR_UNSUPPORTED; }

int rsis_comm_unique_id(void* id_out) {
  if (!id_out) return RSIS_ERR_ARG;
  if (!load_rccl()) return RSIS_ERR_UNSUPPORTED;
  nccl_uid id;
  const int rc = g_rccl.get_uid(&id);
  if (rc != 0) return fail("ncclGetUniqueId", rc);
  memcpy(id_out, &id, sizeof id);
  return RSIS_OK;
}

int rsis_comm_init(void** comm, int world, int rank, const void* id) {
  if (!comm || !id || world < 1 || rank < 0 || rank >= world) return RSIS_ERR_ARG;
  if (!load_rccl()) return RSIS_ERR_UNSUPPORTED;
  nccl_uid uid;
  memcpy(&uid, id, sizeof uid);
  nccl_comm c = nullptr;
  const int rc = g_rccl.init_rank(&c, world, uid, rank);
  if (rc != 0) return fail("ncclCommInitRank", rc);
  *comm = c;
  return RSIS_OK;
}

int rsis_comm_size(void* comm) {
  if (!comm || !load_rccl()) return -1;
  int n = -1;
  return g_rccl.count((nccl_comm)comm, &n) == 0 ? n : -1;
}

int rsis_comm_allreduce_sum_f32(void* comm, float* buf, long n, void* stream) {
  if (!comm || !buf || n < 1) return RSIS_ERR_ARG;
  if (!load_rccl()) return RSIS_ERR_UNSUPPORTED;
  const int rc = g_rccl.allreduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, (nccl_comm)comm, (hipStream_t)stream);
  return rc == 0 ? RSIS_OK : fail("ncclAllReduce", rc);
}

int rsis_comm_destroy(void* comm) {
  if (!comm) return RSIS_OK;
  if (!load_rccl()) return RSIS_ERR_UNSUPPORTED;
  const int rc = g_rccl.destroy((nccl_comm)comm);
  return rc == 0 ? RSIS_OK : fail("ncclCommDestroy", rc);
}
}
