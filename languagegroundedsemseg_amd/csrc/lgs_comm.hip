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
#include <stdlib.h>
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

// ---- device-side mailbox exchange (knob SYNCBN_IPC; round 5).  Every rank owns ONE mailbox in its HBM
//   float    slot[kMbRing][world][kMbRec]    the record rank r contributed to collective number seq, in slot seq % kMbRing
//   uint32_t flag[kMbRing][world]            = seq once that record is complete
// and maps every peer's mailbox through hipIpcGetMemHandle / hipIpcOpenMemHandle.  A SyncBN exchange is then ONE kernel of one
// workgroup on the compute stream: write my record into every rank's mailbox (peer-to-peer stores), fence, raise my flag there,
// spin on the flags of MY mailbox until all `world` records of this seq are in, combine.  No RCCL kernel (15 - 20 us each, 124
// per step), no second communicator next to ProcessGroupNCCL's.  A rank can run at most one collective ahead of the slowest
// rank (the next one needs everybody's record of this one), so a ring of kMbRing = 4 slots is never overwritten while unread.
// What one GPU cannot show: that a peer's stores become visible to a SPINNING kernel across xGMI (the mailbox is fine-grained
// memory when the runtime grants it, flags are system-scope atomics, records are read with system-scope loads); the logic --
// ring, sequence numbers, order of the combination -- is exercised by two processes sharing one GPU (tests/test_gpu_syncbn.py).
constexpr int kMbRing = 4, kMbRec = 2 * 2048 + 1, kMbMaxWorld = 16;
struct Mailbox {
  float *slot;          // [kMbRing][world][kMbRec]
  unsigned *flag;       // [kMbRing][world]
};

struct lgs_comm {
  nccl_comm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  // mailbox mode
  bool ipc = false;
  void *mine = nullptr;                   // this rank's mailbox allocation
  void *peer[kMbMaxWorld] = {nullptr};    // mapped mailboxes of all ranks (peer[rank] == mine)
  Mailbox *d_boxes = nullptr;             // device array [world]
  unsigned seq = 0;                       // collectives issued so far (identical on every rank: same model, same order)
  unsigned *h_err = nullptr;              // pinned, device-mapped word: a kernel that gave up waiting stores its seq here
  unsigned *d_err = nullptr;              // device view of h_err
  long long timeout_ticks = 0;            // wall_clock64() ticks a kernel waits for its peers before it gives up
};

namespace {
inline size_t mb_slot_bytes(int world) { return (size_t)kMbRing * world * kMbRec * sizeof(float); }
inline size_t mb_bytes(int world) { return mb_slot_bytes(world) + (size_t)kMbRing * world * sizeof(unsigned) + 256; }

// The wait is BOUNDED (advisor, round 5): ranks whose sequence numbers diverge -- one rank in eval mode, a launch that failed after
// seq was bumped -- would otherwise spin on the compute stream for ever.  A waiter that sees no record for `timeout` wall-clock
// ticks stores seq into the communicator's host-mapped error word and leaves (its result is garbage); the host side reads that
// word at the head of every sync call and fails with a message instead of hanging.
__device__ inline void mb_put_and_wait(const Mailbox *boxes, int world, int rank, unsigned seq, const float *rec, int len,
                                       unsigned *err, long long timeout) {
  const int ring = (int)(seq % kMbRing);
  // my record into every rank's mailbox (mine included)
  for (int p = 0; p < world; ++p) {
    float *dst = boxes[p].slot + ((size_t)ring * world + rank) * kMbRec;
    for (int i = threadIdx.x; i < len; i += blockDim.x) __hip_atomic_store(dst + i, rec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < world)
    __hip_atomic_store(boxes[threadIdx.x].flag + ring * world + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // everybody's record of THIS seq in my mailbox
  if ((int)threadIdx.x < world) {
    const unsigned *f = boxes[rank].flag + ring * world + threadIdx.x;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 1023u) == 0 && timeout > 0 && wall_clock64() - t0 > timeout) {
        __hip_atomic_store(err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();
}
__device__ inline float mb_get(const Mailbox *boxes, int world, int rank, unsigned seq, int r, int i) {
  const float *src = boxes[rank].slot + ((size_t)(seq % kMbRing) * world + r) * kMbRec;
  return __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// all-gather of the [mean | M2 | count] records: all[r][0 .. 2C] for every rank r, in rank order
__global__ __launch_bounds__(256) void k_mbox_allgather(const Mailbox *boxes, int world, int rank, unsigned seq, const float *local, int len,
                                                        float *all, unsigned *err, long long timeout) {
  mb_put_and_wait(boxes, world, rank, seq, local, len, err, timeout);
  for (int r = 0; r < world; ++r)
    for (int i = threadIdx.x; i < len; i += blockDim.x) all[(size_t)r * len + i] = mb_get(boxes, world, rank, seq, r, i);
}
// all-reduce (sum, fixed rank order: identical bits on every rank) of `len` floats, in place
__global__ __launch_bounds__(256) void k_mbox_allreduce(const Mailbox *boxes, int world, int rank, unsigned seq, float *sums, int len,
                                                        unsigned *err, long long timeout) {
  mb_put_and_wait(boxes, world, rank, seq, sums, len, err, timeout);
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < world; ++r) a += mb_get(boxes, world, rank, seq, r, i);
    sums[i] = a;
  }
}
// next sequence number of a mailbox exchange; fails (instead of queueing more kernels that would wait for the same peers) once a
// kernel has reported a timeout.  0 is never used: the flags are zero-initialised, a wrapped-around seq 0 would be "already there"
inline int mbox_next(lgs_comm *c) {
  if (c->h_err && *reinterpret_cast<volatile unsigned *>(c->h_err) != 0u) {
    lgs::set_error("SyncBN mailbox exchange " + std::to_string(*reinterpret_cast<volatile unsigned *>(c->h_err)) + " timed out waiting for a peer's "
                   "record: the ranks' collective sequences have diverged (a rank in eval mode, a failed launch) or a peer died");
    return 3;
  }
  c->seq += 1;
  if (c->seq == 0) c->seq = 1;
  return 0;
}
}  // namespace

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

// ---- mailbox mode: create (allocates this rank's mailbox) -> exchange the 64-byte handles out of band -> open the peers'
int lgs_comm_create_ipc(int world, int rank, int device, lgs_comm **out, void *handle64) {
  LGS_REQUIRE(out && handle64 && world >= 1 && world <= kMbMaxWorld && rank >= 0 && rank < world, "lgs_comm_create_ipc: bad argument");
  LGS_HIP(hipSetDevice(device));
  lgs_comm *c = new lgs_comm();
  c->world = world; c->rank = rank; c->device = device; c->ipc = true;
  const size_t bytes = mb_bytes(world);
  // fine-grained (coherent with peers while a kernel runs) when the runtime grants it for an IPC-exportable allocation
  hipIpcMemHandle_t h;
  bool ok = hipExtMallocWithFlags(&c->mine, bytes, hipDeviceMallocFinegrained) == hipSuccess && hipIpcGetMemHandle(&h, c->mine) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    if (c->mine) (void)hipFree(c->mine);
    c->mine = nullptr;
    // an ordinary (coarse-grained) allocation gives no guarantee that a peer DEVICE's stores become visible to a kernel that is
    // already spinning: that is "mailbox unavailable" (every rank then agrees to keep the collectives, ddp.EngineComm), unless the
    // caller says all ranks share one device (LGS_MBOX_ALLOW_COARSE=1: the single-GPU logic tests)
    const char *allow = getenv("LGS_MBOX_ALLOW_COARSE");
    if (!(allow && atoi(allow) != 0)) {
      lgs::set_error("lgs_comm_create_ipc: the runtime grants no fine-grained, IPC-exportable allocation for the mailbox; a coarse-grained "
                     "one is not coherent with a spinning kernel across devices (LGS_MBOX_ALLOW_COARSE=1 accepts it when all ranks share one GPU)");
      delete c;
      return 3;
    }
    if (hipMalloc(&c->mine, bytes) != hipSuccess || hipIpcGetMemHandle(&h, c->mine) != hipSuccess) {
      lgs::set_error("lgs_comm_create_ipc: could not allocate / export the mailbox (hipIpcGetMemHandle; HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
      if (c->mine) (void)hipFree(c->mine);
      delete c;
      return 3;
    }
  }
  LGS_HIP(hipMemset(c->mine, 0, bytes));
  LGS_HIP(hipDeviceSynchronize());
  LGS_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_err), sizeof(unsigned), hipHostMallocMapped));
  *c->h_err = 0;
  LGS_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_err), c->h_err, 0));
  {
    int khz = 100000;                                       // wall_clock64() rate; 100 MHz on every CDNA part so far
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device);
    const char *t = getenv("LGS_MBOX_TIMEOUT_S");
    const double secs = t ? atof(t) : 30.0;
    c->timeout_ticks = (long long)(secs * 1000.0 * (double)(khz > 0 ? khz : 100000));
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI hands the handle over as 64 bytes");
  ::memcpy(handle64, &h, 64);
  c->peer[rank] = c->mine;
  *out = c;
  return 0;
}

int lgs_comm_ipc_open(lgs_comm *c, const void *handles64) {
  LGS_REQUIRE(c && c->ipc && handles64, "lgs_comm_ipc_open: bad argument");
  LGS_HIP(hipSetDevice(c->device));
  Mailbox host[kMbMaxWorld];
  for (int r = 0; r < c->world; ++r) {
    if (r != c->rank) {
      hipIpcMemHandle_t h;
      ::memcpy(&h, reinterpret_cast<const char *>(handles64) + 64 * r, 64);
      LGS_HIP(hipIpcOpenMemHandle(&c->peer[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    host[r].slot = reinterpret_cast<float *>(c->peer[r]);
    host[r].flag = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(c->peer[r]) + mb_slot_bytes(c->world));
  }
  LGS_HIP(hipMalloc(&c->d_boxes, sizeof(Mailbox) * c->world));
  LGS_HIP(hipMemcpy(c->d_boxes, host, sizeof(Mailbox) * c->world, hipMemcpyHostToDevice));
  c->seq = 0;
  return 0;
}

int lgs_comm_destroy(lgs_comm *c) {
  if (!c) return 0;
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  if (c->ipc) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
      if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->d_boxes) (void)hipFree(c->d_boxes);
    if (c->mine) (void)hipFree(c->mine);
    if (c->h_err) (void)hipHostFree(c->h_err);
  }
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
  LGS_REQUIRE(comm && (comm->comm || (comm->ipc && comm->d_boxes)) && x && y && gamma && beta && stats && inv_n && workspace,
              "lgs_bn_forward_sync: null argument");
  char *ws = reinterpret_cast<char *>(workspace);
  float *local = reinterpret_cast<float *>(ws + align256(lgs_bn_workspace_bytes(n, c)));
  float *all = local + (2 * c + 1);
  int rc;
  if ((rc = lgs_bn_stats(x, n, c, local, dtype, workspace, nullptr, 0, nullptr, stream))) return rc;
  if (comm->ipc) {
    LGS_REQUIRE(2 * c + 1 <= kMbRec, "lgs_bn_forward_sync: record wider than a mailbox slot");
    if (mbox_next(comm)) return 3;
    LGS_KLAUNCH(k_mbox_allgather, 1, 256, 0, (hipStream_t)stream, comm->d_boxes, comm->world, comm->rank, comm->seq, local, 2 * c + 1, all,
                comm->d_err, comm->timeout_ticks);
    LGS_HIP(hipGetLastError());
  } else {
    LGS_NCCL(rccl().AllGather(local, all, (size_t)(2 * c + 1), kNcclFloat32, comm->comm, (hipStream_t)stream));
  }
  if ((rc = lgs_bn_sync_combine(all, comm->world, c, eps, momentum, running_mean, running_var, num_batches_tracked, stats, inv_n, stream)))
    return rc;
  return lgs_bn_apply(x, n, c, gamma, beta, stats, residual, relu, y, dtype, y_row_stride, stream);
}

// backward: local [sum dy' | sum dy' xhat] (also the parameter gradients, which stay local) -> all-reduce -> apply
int lgs_bn_backward_sync(lgs_comm *comm, const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                         const float *beta, const float *stats, const float *inv_n, int relu, void *dx, void *dresidual, float *dgamma,
                         float *dbeta, int dtype, void *workspace, int64_t dy_row_stride, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(comm && (comm->comm || (comm->ipc && comm->d_boxes)) && x && dy && dx && gamma && stats && inv_n && workspace,
              "lgs_bn_backward_sync: null argument");
  char *ws = reinterpret_cast<char *>(workspace);
  float *sums = reinterpret_cast<float *>(ws + align256(lgs_bn_workspace_bytes(n, c)) + align256((int64_t)(comm->world + 1) * (2 * c + 1) * 4));
  int rc;
  if ((rc = lgs_bn_backward_reduce(x, y, dy, n, c, gamma, beta, stats, relu, sums, dgamma, dbeta, dtype, workspace, dy_row_stride, y_row_stride,
                                   stream)))
    return rc;
  if (comm->ipc) {
    if (mbox_next(comm)) return 3;
    LGS_KLAUNCH(k_mbox_allreduce, 1, 256, 0, (hipStream_t)stream, comm->d_boxes, comm->world, comm->rank, comm->seq, sums, 2 * c,
                comm->d_err, comm->timeout_ticks);
    LGS_HIP(hipGetLastError());
  } else {
    LGS_NCCL(rccl().AllReduce(sums, sums, (size_t)(2 * c), kNcclFloat32, kNcclSum, comm->comm, (hipStream_t)stream));
  }
  return lgs_bn_backward_apply(x, y, dy, n, c, gamma, beta, stats, sums, 0.f, inv_n, relu, dx, dresidual, dtype, dy_row_stride, y_row_stride, stream);
}

}  // extern "C"
