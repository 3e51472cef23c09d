// PointGroup clustering on the device (validation-time only in the reference).
//
// Replaces   PG_OP.ballquery_batch_p  /root/reference/downstream/insseg/lib/bfs/ops/src/bfs_cluster_kernel.cu:16-61
//            (per point an O(n) brute-force scan of its batch segment, d2 < r2, <= 1000 hits, atomicAdd cursor)
//   and      PG_OP.bfs_cluster        .../bfs_cluster.cpp:54-125 (CPU std::queue BFS over same-label neighbours)
//   as used by lib/bfs/bfs.py:124-150 (Clustering.cluster_).
//
// MI355X design: points are binned into cells of edge `radius` (64-bit cell key, one radix sort); a point's neighbours
// within `radius` can only live in the 27 surrounding cells, each found by binary search in the sorted key array -- the
// O(n^2) scan becomes O(n * 27 * (log n + cell occupancy)).  Connected components of the graph
// {(i, j): d2(i, j) < r2, batch equal, semantic label equal} are found with a lock-free union-find (roots always link
// towards the smaller index, so a component's representative is its smallest point index -- exactly the point the
// reference's BFS starts the component from, which makes the cluster ORDER identical), then components smaller than
// `threshold` are dropped.  No neighbour lists are materialised.
//
// d2 is evaluated in float32 with explicitly rounded operations in the reference's order
// (ox-x)^2 + (oy-y)^2 + (oz-z)^2 (no FMA contraction), so membership is bit-defined and equals the oracle's.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <string>

#include "../../include/lgs_engine.h"
#include "lgs_common.h"

namespace lgs {

constexpr int kCellBias = 1 << 17;   // |cell| < 2^17 per axis (18 bits each), batch < 1024 in the top bits

__device__ inline uint64_t cell_key(int b, int cx, int cy, int cz) {
  return ((uint64_t)(uint32_t)b << 54) | ((uint64_t)(uint32_t)(cx + kCellBias) << 36) |
         ((uint64_t)(uint32_t)(cy + kCellBias) << 18) | (uint64_t)(uint32_t)(cz + kCellBias);
}

__global__ void k_cell_keys(const float *__restrict__ xyz, const int32_t *__restrict__ batch, int64_t n, float inv_r,
                            uint64_t *__restrict__ keys, int32_t *__restrict__ vals, int *__restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cx = (int)floorf(xyz[3 * i] * inv_r), cy = (int)floorf(xyz[3 * i + 1] * inv_r), cz = (int)floorf(xyz[3 * i + 2] * inv_r);
  const int b = batch ? batch[i] : 0;
  const int lim = kCellBias - 2;
  if (b < 0 || b >= 1024 || cx <= -lim || cx >= lim || cy <= -lim || cy >= lim || cz <= -lim || cz >= lim) {
    atomicOr(err, 1);
    keys[i] = ~0ull;
  } else {
    keys[i] = cell_key(b, cx, cy, cz);
  }
  vals[i] = (int32_t)i;
}

__device__ inline int64_t lower_bound_key(const uint64_t *__restrict__ skeys, int64_t n, uint64_t k) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skeys[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ inline int32_t uf_find(int32_t *parent_, int32_t i) {
  volatile int32_t *parent = parent_;   // other threads shorten paths concurrently; pointers only ever move towards the root
  while (true) {
    const int32_t p = parent[i];
    if (p == i) return i;
    const int32_t gp = parent[p];
    if (gp != p) parent[i] = gp;        // path halving (benign race: gp is an ancestor of i either way)
    i = p;
  }
}

__device__ inline void uf_union(int32_t *parent, int32_t a, int32_t b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    const int32_t hi = a > b ? a : b, lo = a > b ? b : a;
    const int32_t old = atomicCAS(&parent[hi], hi, lo);     // the larger root links to the smaller one
    if (old == hi) return;
    a = old; b = lo;
  }
}

// one thread per point (in sorted-by-cell order, so neighbouring threads probe the same cells)
__global__ void k_union_neighbours(const float *__restrict__ xyz, const int32_t *__restrict__ batch, const int32_t *__restrict__ sem,
                                   const uint64_t *__restrict__ skeys, const int32_t *__restrict__ svals, int64_t n, float inv_r,
                                   float r2, int32_t *__restrict__ parent) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (skeys[s] == ~0ull) return;
  const int32_t i = svals[s];
  const float ox = xyz[3 * (int64_t)i], oy = xyz[3 * (int64_t)i + 1], oz = xyz[3 * (int64_t)i + 2];
  const int cx = (int)floorf(ox * inv_r), cy = (int)floorf(oy * inv_r), cz = (int)floorf(oz * inv_r);
  const int b = batch ? batch[i] : 0;
  const int32_t li = sem[i];
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy) {
      // the three z-cells of a (dx, dy) column are consecutive keys: one search, one contiguous scan
      const uint64_t k0 = cell_key(b, cx + dx, cy + dy, cz - 1), k1 = cell_key(b, cx + dx, cy + dy, cz + 1);
      for (int64_t p = lower_bound_key(skeys, n, k0); p < n && skeys[p] <= k1; ++p) {
        const int32_t j = svals[p];
        if (j >= i || sem[j] != li) continue;               // every edge once, from its larger endpoint
        const float ddx = __fsub_rn(ox, xyz[3 * (int64_t)j]), ddy = __fsub_rn(oy, xyz[3 * (int64_t)j + 1]),
                    ddz = __fsub_rn(oz, xyz[3 * (int64_t)j + 2]);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz));
        if (d2 < r2) uf_union(parent, i, j);
      }
    }
}

__global__ void k_iota(int32_t *p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}
__global__ void k_flatten_count(int32_t *__restrict__ parent, int64_t n, int32_t *__restrict__ size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = uf_find(parent, (int32_t)i);
  parent[i] = r;
  atomicAdd(&size[r], 1);
}
__global__ void k_component_out(const int32_t *__restrict__ parent, const int32_t *__restrict__ size, int64_t n, int threshold,
                                int32_t *__restrict__ component, int32_t *__restrict__ n_clusters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = parent[i];
  const bool keep = size[r] >= threshold;
  component[i] = keep ? r : -1;
  if (keep && r == (int32_t)i) atomicAdd(n_clusters, 1);
}

}  // namespace lgs

using namespace lgs;

extern "C" {

int64_t lgs_cluster_workspace_bytes(int64_t n) {
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                                  (size_t)(n > 0 ? n : 1), 0, 64, (hipStream_t)0);
  return align256((int64_t)tb) + 2 * align256(8 * (n + 1)) + 4 * align256(4 * (n + 1)) + 512;
}

int lgs_cluster(const float *xyz, const int32_t *batch_idx, const int32_t *semantic_label, int64_t n, float radius, int threshold,
                int32_t *component, int32_t *n_clusters, void *workspace, void *stream) {
  LGS_REQUIRE(n >= 0 && n < (1ll << 31) - 1 && radius > 0.f && n_clusters, "lgs_cluster: bad argument");
  *n_clusters = 0;
  if (n == 0) return 0;
  LGS_REQUIRE(xyz && semantic_label && component && workspace, "lgs_cluster: null argument");
  hipStream_t s = (hipStream_t)stream;
  char *ws = reinterpret_cast<char *>(workspace);
  size_t tb = 0;
  LGS_HIP(rocprim::radix_sort_pairs(nullptr, tb, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                                    (size_t)n, 0, 64, s));
  void *tmp = ws; ws += align256((int64_t)tb);
  uint64_t *keys = reinterpret_cast<uint64_t *>(ws); ws += align256(8 * (n + 1));
  uint64_t *skeys = reinterpret_cast<uint64_t *>(ws); ws += align256(8 * (n + 1));
  int32_t *vals = reinterpret_cast<int32_t *>(ws); ws += align256(4 * (n + 1));
  int32_t *svals = reinterpret_cast<int32_t *>(ws); ws += align256(4 * (n + 1));
  int32_t *parent = reinterpret_cast<int32_t *>(ws); ws += align256(4 * (n + 1));
  int32_t *size = reinterpret_cast<int32_t *>(ws); ws += align256(4 * (n + 1));
  int32_t *scal = reinterpret_cast<int32_t *>(ws);   // [0] error flag, [1] cluster count
  const unsigned nb = (unsigned)((n + 255) / 256);
  // cells a hair larger than the radius: two points closer than `radius` then provably sit in adjacent cells even after
  // the rounding of x * inv_r
  const float inv_r = 1.0f / (radius * 1.0001f), r2 = radius * radius;
  LGS_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(int32_t), s));
  LGS_HIP(hipMemsetAsync(size, 0, sizeof(int32_t) * n, s));
  LGS_KLAUNCH(k_cell_keys, nb, 256, 0, s, xyz, batch_idx, n, inv_r, keys, vals, scal);
  LGS_HIP(rocprim::radix_sort_pairs(tmp, tb, keys, skeys, vals, svals, (size_t)n, 0, 64, s));
  LGS_KLAUNCH(k_iota, nb, 256, 0, s, parent, n);
  LGS_KLAUNCH(k_union_neighbours, nb, 256, 0, s, xyz, batch_idx, semantic_label, skeys, svals, n, inv_r, r2, parent);
  LGS_KLAUNCH(k_flatten_count, nb, 256, 0, s, parent, n, size);
  LGS_KLAUNCH(k_component_out, nb, 256, 0, s, parent, size, n, threshold, component, scal + 1);
  int32_t h[2] = {0, 0};
  LGS_HIP(hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, s));
  LGS_HIP(hipStreamSynchronize(s));
  LGS_REQUIRE(h[0] == 0, "lgs_cluster: a point lies outside the supported cell range (|x / radius| < 131070, batch < 1024)");
  *n_clusters = h[1];
  return 0;
}

}  // extern "C"
