// lgs_comm.hip -- the engine's own RCCL communicator and SyncBatchNorm as ONE call per direction.
//
// MinkowskiSyncBatchNorm (ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm, /root/reference/main.py:121-123; per layer one exchange
// of batch statistics forward and one of the two gradient sums backward, as torch.nn.SyncBatchNorm does) was five engine calls and
// two torch.distributed collectives per layer and direction.  Through ProcessGroupNCCL every one of the 124 small collectives of a
// step costs ~60 us of host time (Python, c10d work objects) and two stream hand-overs (compute stream -> the process group's
// stream and back: ~20-30 us of compute-stream idle each) -- measured on one GPU with a world of one rank (bench.py
// `dp_path_world1`): forward of the 8-scene step 13.0 ms, host-bound, against 10.7 ms without SyncBN.
// Here the engine holds an RCCL communicator of its own (created from an id the ranks exchange once through torch.distributed) and
// issues ncclAllGather / ncclAllReduce ON THE COMPUTE STREAM between its kernels: statistics -> all-gather -> combine -> apply is one
// host call, no stream switch, no Python in between.  RCCL is resolved at run time (dlopen of the librccl the process already
// has), so the engine library keeps no link-time dependency on it.
#include "lgs_common.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

// the subset of rccl.h this file uses (kept local: the public header stays free of RCCL types)
typedef struct { char internal[128]; } nccl_unique_id;
typedef void *nccl_comm_t;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };
struct Rccl {
  int (*GetUniqueId)(nccl_unique_id *) = nullptr;
  int (*CommInitRank)(nccl_comm_t *, int, nccl_unique_id, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the copy the process already loaded (torch ships one and links it), else the system's
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!h)
      for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce;
    if (!r.ok) r.why = "librccl lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllGather / AllReduce";
  });
  return r;
}

#define LGS_NCCL(expr)                                                                                         \
  do {                                                                                                         \
    const int _r = (expr);                                                                                     \
    if (_r != kNcclSuccess) {                                                                                  \
      lgs::set_error(std::string(#expr) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(_r) : "?") + \
                     " (" __FILE__ ":" + std::to_string(__LINE__) + ")");                                      \
      return 3;                                                                                                \
    }                                                                                                          \
  } while (0)

inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

}  // namespace

struct lgs_comm {
  nccl_comm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
};

extern "C" {

int lgs_comm_unique_id(void *id128) {
  LGS_REQUIRE(id128, "lgs_comm_unique_id: null argument");
  LGS_REQUIRE(rccl().ok, ("lgs_comm_unique_id: " + rccl().why).c_str());
  nccl_unique_id id;
  LGS_NCCL(rccl().GetUniqueId(&id));
  ::memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

int lgs_comm_create(const void *id128, int world, int rank, int device, lgs_comm **out) {
  LGS_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "lgs_comm_create: bad argument");
  LGS_REQUIRE(rccl().ok, ("lgs_comm_create: " + rccl().why).c_str());
  LGS_HIP(hipSetDevice(device));
  nccl_unique_id id;
  ::memcpy(id.internal, id128, sizeof(id.internal));
  lgs_comm *c = new lgs_comm();
  c->world = world; c->rank = rank; c->device = device;
  const int r = rccl().CommInitRank(&c->comm, world, id, rank);
  if (r != kNcclSuccess) {
    lgs::set_error(std::string("ncclCommInitRank failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "?"));
    delete c;
    return 3;
  }
  *out = c;
  return 0;
}

int lgs_comm_destroy(lgs_comm *c) {
  if (!c) return 0;
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  delete c;
  return 0;
}

int lgs_comm_world(const lgs_comm *c) { return c ? c->world : 0; }

int64_t lgs_bn_sync_workspace_bytes(int64_t n, int c, int world) {
  // BatchNorm scratch | local record [2C+1] | records of all ranks [world][2C+1] | gradient sums [2C]
  return align256(lgs_bn_workspace_bytes(n, c)) + align256((int64_t)(world + 1) * (2 * c + 1) * 4) + align256(2 * c * 4) + 256;
}

// forward: local (mean, M2, count) -> all-gather of the records -> Chan's combination (+ running statistics) -> normalise
// (+ residual) (+ ReLU); stats [2C] and inv_n [1] (device: 1 / global rows) are what the backward needs
int lgs_bn_forward_sync(lgs_comm *comm, const void *x, int64_t n, int c, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, int64_t *num_batches_tracked, const void *residual,
                        int relu, void *y, float *stats, float *inv_n, int dtype, void *workspace, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(comm && comm->comm && x && y && gamma && beta && stats && inv_n && workspace, "lgs_bn_forward_sync: null argument");
  char *ws = reinterpret_cast<char *>(workspace);
  float *local = reinterpret_cast<float *>(ws + align256(lgs_bn_workspace_bytes(n, c)));
  float *all = local + (2 * c + 1);
  int rc;
  if ((rc = lgs_bn_stats(x, n, c, local, dtype, workspace, nullptr, 0, nullptr, stream))) return rc;
  LGS_NCCL(rccl().AllGather(local, all, (size_t)(2 * c + 1), kNcclFloat32, comm->comm, (hipStream_t)stream));
  if ((rc = lgs_bn_sync_combine(all, comm->world, c, eps, momentum, running_mean, running_var, num_batches_tracked, stats, inv_n, stream)))
    return rc;
  return lgs_bn_apply(x, n, c, gamma, beta, stats, residual, relu, y, dtype, y_row_stride, stream);
}

// backward: local [sum dy' | sum dy' xhat] (also the parameter gradients, which stay local) -> all-reduce -> apply
int lgs_bn_backward_sync(lgs_comm *comm, const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                         const float *beta, const float *stats, const float *inv_n, int relu, void *dx, void *dresidual, float *dgamma,
                         float *dbeta, int dtype, void *workspace, int64_t dy_row_stride, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(comm && comm->comm && x && dy && dx && gamma && stats && inv_n && workspace, "lgs_bn_backward_sync: null argument");
  char *ws = reinterpret_cast<char *>(workspace);
  float *sums = reinterpret_cast<float *>(ws + align256(lgs_bn_workspace_bytes(n, c)) + align256((int64_t)(comm->world + 1) * (2 * c + 1) * 4));
  int rc;
  if ((rc = lgs_bn_backward_reduce(x, y, dy, n, c, gamma, beta, stats, relu, sums, dgamma, dbeta, dtype, workspace, dy_row_stride, y_row_stride,
                                   stream)))
    return rc;
  LGS_NCCL(rccl().AllReduce(sums, sums, (size_t)(2 * c), kNcclFloat32, kNcclSum, comm->comm, (hipStream_t)stream));
  return lgs_bn_backward_apply(x, y, dy, n, c, gamma, beta, stats, sums, 0.f, inv_n, relu, dx, dresidual, dtype, dy_row_stride, y_row_stride, stream);
}

}  // extern "C"
