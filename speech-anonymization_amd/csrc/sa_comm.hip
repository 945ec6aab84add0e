// sa_comm.hip -- the data-parallel exchange of the train step: one RCCL communicator, one side
// stream and two events per process, owned by the library (include/sa_hip.h
// "data-parallel exchange").  Replaces, for this path, what speechbrain's Brain does for the
// reference with DistributedDataParallel / SyncBatchNorm after ddp_init_group
// (speechbrain_convae_train.py:524): the gradient average of the three stage buckets and the
// BatchNorm statistic sums.
//
// Stream discipline (host code only, no kernels here):
//   sa_comm_allreduce(buf, ..., producer):  record ev_in on `producer`, side waits ev_in,
//       ncclAllReduce in place on the side stream.  Returns at once; the producer stream goes on
//       with the rest of backward while the collective (latency-bound at 0.6-0.9 MB) runs.
//   sa_comm_allreduce_inline(buf, ..., stream):  the collective in `stream` itself, no event hop:
//       for the BatchNorm statistic sums, whose producer is the previous kernel and whose consumer
//       the next one (measured on one rank: two event hops per exchange cost ~35 us of bubble
//       each, twelve times per step).
//   sa_comm_join(consumer):                 record ev_out on side, `consumer` waits ev_out.
//       Called once per backward, before the optimizer reads the gradients.
// RCCL is bound at run time (dlopen "librccl.so.1"): the process usually holds torch's copy
// already, and two copies of RCCL in one process must not happen; the library has no link-time
// dependency on it, and single-process use never touches it.
#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/sa_hip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
};

struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  int rank = 0, world = 0, device = -1;
  long long ncalls = 0;
};

Rccl g_rccl;
Comm g_comm;

int nccl_rc(ncclResult_t r) { return r == ncclSuccess ? 0 : -(1000 + (int)r); }
int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }

int bind_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return -ENOSYS;
  Rccl r;
  r.handle = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) return -ENOSYS;
  g_rccl = r;
  return 0;
}

}  // namespace

extern "C" {

int sa_comm_unique_id(void* id128) {
  if (!id128) return -EINVAL;
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  if (int rc = nccl_rc(g_rccl.GetUniqueId(&id))) return rc;
  static_assert(sizeof(id) == SA_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int sa_comm_init(int rank, int world, const void* id128, int device) {
  if (!id128 || world < 1 || rank < 0 || rank >= world || device < 0) return -EINVAL;
  if (g_comm.comm) return -EEXIST;
  if (int rc = bind_rccl()) return rc;
  if (int rc = hip_rc(hipSetDevice(device))) return rc;
  Comm c;
  c.rank = rank;
  c.world = world;
  c.device = device;
  int rc = hip_rc(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
  if (!rc) rc = hip_rc(hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming));
  if (!rc) rc = hip_rc(hipEventCreateWithFlags(&c.ev_out, hipEventDisableTiming));
  if (!rc) {
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    rc = nccl_rc(g_rccl.CommInitRank(&c.comm, world, id, rank));
  }
  if (rc) {
    if (c.ev_out) (void)hipEventDestroy(c.ev_out);
    if (c.ev_in) (void)hipEventDestroy(c.ev_in);
    if (c.side) (void)hipStreamDestroy(c.side);
    return rc;
  }
  g_comm = c;
  return 0;
}

int sa_comm_world(void) { return g_comm.comm ? g_comm.world : 0; }

int sa_comm_allreduce(void* buf, long long n, int dtype, int avg, void* producer_stream) {
  if (!g_comm.comm) return -ENOTCONN;
  if (!buf || n <= 0) return -EINVAL;
  ncclDataType_t dt;
  if (dtype == SA_F32) dt = ncclFloat32;
  else if (dtype == SA_F64) dt = ncclFloat64;
  else return -EINVAL;
  hipStream_t prod = (hipStream_t)producer_stream;
  if (int rc = hip_rc(hipEventRecord(g_comm.ev_in, prod))) return rc;
  if (int rc = hip_rc(hipStreamWaitEvent(g_comm.side, g_comm.ev_in, 0))) return rc;
  g_comm.ncalls++;
  return nccl_rc(g_rccl.AllReduce(buf, buf, (size_t)n, dt, avg ? ncclAvg : ncclSum, g_comm.comm,
                                  g_comm.side));
}

int sa_comm_allreduce_inline(void* buf, long long n, int dtype, int avg, void* stream) {
  if (!g_comm.comm) return -ENOTCONN;
  if (!buf || n <= 0) return -EINVAL;
  ncclDataType_t dt;
  if (dtype == SA_F32) dt = ncclFloat32;
  else if (dtype == SA_F64) dt = ncclFloat64;
  else return -EINVAL;
  g_comm.ncalls++;
  return nccl_rc(g_rccl.AllReduce(buf, buf, (size_t)n, dt, avg ? ncclAvg : ncclSum, g_comm.comm,
                                  (hipStream_t)stream));
}

int sa_comm_join(void* consumer_stream) {
  if (!g_comm.comm) return -ENOTCONN;
  if (int rc = hip_rc(hipEventRecord(g_comm.ev_out, g_comm.side))) return rc;
  return hip_rc(hipStreamWaitEvent((hipStream_t)consumer_stream, g_comm.ev_out, 0));
}

int sa_comm_ncalls(void) { return (int)(g_comm.ncalls & 0x7fffffff); }

int sa_comm_destroy(void) {
  if (!g_comm.comm) return 0;
  int rc = hip_rc(hipStreamSynchronize(g_comm.side));
  int r2 = nccl_rc(g_rccl.CommDestroy(g_comm.comm));
  (void)hipEventDestroy(g_comm.ev_in);
  (void)hipEventDestroy(g_comm.ev_out);
  (void)hipStreamDestroy(g_comm.side);
  g_comm = Comm();
  return rc ? rc : r2;
}

}  // extern "C"
