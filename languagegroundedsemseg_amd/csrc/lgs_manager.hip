// lgs_manager.hip -- coordinate manager + kernel-map construction for gfx950.
//
// Replaces (functionally) MinkowskiEngine's CoordinateManager as used by the reference:
//   SparseTensor(feats, coords)            /root/reference/lib/train_test/pl_BaselineTrainer.py:300
//   stride-2 output maps                    /root/reference/models/res16unet.py:49-107
//   kernel maps for k in {1,2,3}            /root/reference/models/modules/common.py:179-236
//
// Design: every coordinate is packed into ONE 64-bit Morton key
//   key = batch << 54 | interleave3(x + 2^17, y + 2^17, z + 2^17)
// Level-0 rows keep the caller's order (logits stay row-aligned with the input), but every map also
// carries its rows in Morton order (`order`: sorted position -> row).  Consequences:
//   * coarsening by 2 is a bit-mask on the key and preserves the sort, so a strided map and its
//     2x2x2 kernel map come from one flag+scan over the sorted keys -- no hashing;
//   * 3x3x3 maps come from 27 probes per voxel into an open-addressing hash (64-bit atomicCAS build);
//   * conv tiles are runs of 64 Morton-consecutive voxels, so a wavefront ballot gives the per-tile
//     bitmask of kernel offsets that have any neighbour at all (planar surfaces miss most
//     out-of-plane offsets), which the conv kernels use to skip work.
#include "lgs_common.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace lgs {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

constexpr int kCoordBits = 18;
constexpr int kBias = 1 << (kCoordBits - 1);  // 131072
constexpr uint64_t kEmpty = ~0ull;

__host__ __device__ inline uint64_t spread3(uint64_t x) {
  x &= 0x1fffff;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__host__ __device__ inline uint32_t compact3(uint64_t x) {
  x &= 0x1249249249249249ull;
  x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
  x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
  x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
  x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
  x = (x ^ (x >> 32)) & 0x1fffff;
  return (uint32_t)x;
}
__device__ inline uint64_t pack_key(int b, int xb, int yb, int zb) {  // biased coords
  return ((uint64_t)b << 54) | spread3((uint64_t)xb) | (spread3((uint64_t)yb) << 1) | (spread3((uint64_t)zb) << 2);
}
__device__ inline void unpack_key(uint64_t key, int &b, int &xb, int &yb, int &zb) {
  b = (int)(key >> 54);
  uint64_t m = key & ((1ull << 54) - 1);
  xb = (int)compact3(m);
  yb = (int)compact3(m >> 1);
  zb = (int)compact3(m >> 2);
}
__device__ inline uint64_t hash64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

// ------------------------------------------------------------------------------------------- kernels
__global__ void k_pack_keys(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals, int *err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = reinterpret_cast<const int4 *>(coords)[i];
  bool ok = c.x >= 0 && c.x < 1024 && c.y > -kBias && c.y < kBias - 64 && c.z > -kBias && c.z < kBias - 64 &&
            c.w > -kBias && c.w < kBias - 64;
  if (!ok) { atomicOr(err, 1); c = make_int4(0, 0, 0, 0); }
  keys[i] = pack_key(c.x, c.y + kBias, c.z + kBias, c.w + kBias);
  vals[i] = (int32_t)i;
}

// head[p] = 1 if sorted key p starts a new run of (key & keep_mask)
__global__ void k_heads(const uint64_t *skeys, int64_t n, uint64_t keep_mask, int32_t *head) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint64_t k = skeys[p] & keep_mask;
  head[p] = (p == 0 || (skeys[p - 1] & keep_mask) != k) ? 1 : 0;
}

// insert step: flag (by input index) the first occurrence of every distinct coordinate
__global__ void k_mark_first(const int32_t *svals, const int32_t *head, int64_t n, int32_t *is_first) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  is_first[svals[p]] = head[p];
}
// urow = exclusive scan of is_first over input order; write coords/unique_index of surviving rows,
// and the sorted-order arrays of the deduplicated map
__global__ void k_emit_unique(const int32_t *coords, const int32_t *is_first, const int32_t *urow, int64_t n,
                              int32_t *ucoords, int64_t *unique_index) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !is_first[i]) return;
  int32_t u = urow[i];
  reinterpret_cast<int4 *>(ucoords)[u] = reinterpret_cast<const int4 *>(coords)[i];
  if (unique_index) unique_index[u] = i;
}
__global__ void k_emit_sorted(const uint64_t *skeys, const int32_t *svals, const int32_t *head,
                              const int32_t *runid_incl, const int32_t *urow, int64_t n, uint64_t *ukeys,
                              int32_t *order) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !head[p]) return;
  int32_t q = runid_incl[p] - 1;
  ukeys[q] = skeys[p];
  order[q] = urow[svals[p]];
}
__global__ void k_emit_inverse(const int32_t *svals, const int32_t *runid_incl, const int32_t *order, int64_t n,
                               int64_t *inverse) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  inverse[svals[p]] = order[runid_incl[p] - 1];
}

// stride-2: coarse rows are created in Morton order (row == sorted position)
// nc = the row count the coarse arrays were SIZED for (counted at insert time, no synchronisation): a run index beyond it is
// never written, and the last thread compares the two counts (d_err bit 1, lgs_manager_check)
__global__ void k_emit_coarse(const uint64_t *fkeys, const int32_t *head, const int32_t *cidx_incl, int64_t n,
                              uint64_t keep_mask, uint64_t *ckeys, int32_t *ccoords, int32_t *cstart,
                              int32_t *fine_cidx, int64_t nc, int *d_err) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int32_t q = cidx_incl[p] - 1;
  fine_cidx[p] = q;
  if (p == n - 1 && (int64_t)q + 1 != nc && d_err) atomicOr(d_err, 2);
  if (q >= nc) return;
  if (head[p]) {
    uint64_t k = fkeys[p] & keep_mask;
    ckeys[q] = k;
    cstart[q] = (int32_t)p;
    int b, x, y, z;
    unpack_key(k, b, x, y, z);
    reinterpret_cast<int4 *>(ccoords)[q] = make_int4(b, x - kBias, y - kBias, z - kBias);
  }
  if (p == n - 1) cstart[q + 1] = (int32_t)n;
}

__global__ void k_hash_fill(uint64_t *hkeys, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) hkeys[i] = kEmpty;
}
__global__ void k_hash_insert(const uint64_t *skeys, const int32_t *order, int64_t n, uint64_t *hkeys,
                              int32_t *hvals, uint64_t capm1) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint64_t key = skeys[p];
  uint64_t s = hash64(key) & capm1;
  for (;;) {
    unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(hkeys + s), (unsigned long long)kEmpty,
                                        (unsigned long long)key);
    if (prev == kEmpty) { hvals[s] = order ? order[p] : (int32_t)p; return; }
    s = (s + 1) & capm1;  // keys are unique here, no equality case
  }
}

// Row counts of ALL coarser levels from the sorted stride-1 keys: coarsening by 2^L drops the 3 L low Morton bits and keeps the
// order, so the level-L map has one row per run of equal masked keys.  Counting the run heads of every level in the pass that
// already exists for the insert lets lgs_manager_insert return them with ITS host synchronisation: the four
// lgs_manager_stride2 calls of the U-Net then need none (5 host syncs per training step -> 1; duplicates of the input do not
// matter: equal keys stay equal under any mask).
constexpr int kPreLevels = 8;
constexpr int kCountPerThread = 16;
__global__ __launch_bounds__(256) void k_count_levels(const uint64_t *__restrict__ skeys, int64_t n, int32_t *__restrict__ counts) {
  // a workgroup walks 256 x kCountPerThread consecutive keys and adds ONE number per level to the global counters (one
  // atomic per wave and level on eight shared addresses cost 0.67 ms at 1.2 M keys: the atomics serialise at the L2)
  __shared__ int32_t l_cnt[kPreLevels];
  if (threadIdx.x < kPreLevels) l_cnt[threadIdx.x] = 0;
  __syncthreads();
  int32_t c[kPreLevels];
#pragma unroll
  for (int L = 0; L < kPreLevels; ++L) c[L] = 0;
  const int64_t base = (int64_t)blockIdx.x * 256 * kCountPerThread;
#pragma unroll 4
  for (int i = 0; i < kCountPerThread; ++i) {
    const int64_t p = base + (int64_t)i * 256 + threadIdx.x;
    if (p >= n) break;
    const uint64_t k = skeys[p], q = p > 0 ? skeys[p - 1] : ~0ull, d = k ^ q;
#pragma unroll
    for (int L = 1; L <= kPreLevels; ++L) c[L - 1] += (p == 0 || (d >> (3 * L)) != 0) ? 1 : 0;
  }
#pragma unroll
  for (int L = 0; L < kPreLevels; ++L) {
    int32_t v = c[L];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&l_cnt[L], v);
  }
  __syncthreads();
  if (threadIdx.x < kPreLevels && l_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], l_cnt[threadIdx.x]);
}

// 3x3x3 stride-1 map: one thread per sorted position, 27 probes; nbr is offset-major [27][n_pad].
// Round 6: the probes of one z-plane (nine offsets) are issued TOGETHER -- nine hash slots computed, nine key loads in flight, then
// the (rare) continued probes, then nine value loads in flight -- instead of 27 dependent load chains one after the other: the
// kernel is bound by the latency of its random accesses into the 20 - 30 MB table, not by their number (level 0 of the 8-scene
// batch: 0.87 -> see profiles/r06_experiments.txt).  Same table, same probe sequence per offset: the map is bit-identical.
__global__ void k_build_map3(const uint64_t *skeys, int64_t n, int64_t n_pad, int ts, const uint64_t *hkeys,
                             const int32_t *hvals, uint64_t capm1, int32_t *nbr, uint32_t *pmask) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // blockDim multiple of 64; p < n_pad by grid
  bool live = p < n;
  int b = 0, x = 0, y = 0, z = 0;
  if (live) unpack_key(skeys[p], b, x, y, z);
  uint32_t m = 0;
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    uint64_t key[9], slot[9], hk[9];
    bool ok[9];
    const int zz = z + (g - 1) * ts;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int xx = x + (j % 3 - 1) * ts, yy = y + (j / 3 - 1) * ts;
      ok[j] = live && (((unsigned)xx | (unsigned)yy | (unsigned)zz) < (1u << kCoordBits));
      key[j] = ok[j] ? pack_key(b, xx, yy, zz) : 0ull;
      slot[j] = hash64(key[j]) & capm1;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) hk[j] = ok[j] ? hkeys[slot[j]] : kEmpty;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      if (ok[j]) {
        uint64_t cur = hk[j], sl = slot[j];
        while (cur != key[j] && cur != kEmpty) {
          sl = (sl + 1) & capm1;
          cur = hkeys[sl];
        }
        slot[j] = sl;
        ok[j] = cur == key[j];
      }
    }
    int32_t r[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) r[j] = ok[j] ? hvals[slot[j]] : -1;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int k = 9 * g + j;
      nbr[(int64_t)k * n_pad + p] = r[j];
      if (r[j] >= 0) m |= 1u << k;
    }
  }
  pmask[p] = m;
}

// Rows whose neighbourhoods have the same shape are clustered: inside windows of kMaskWindow Morton-consecutive
// positions (spatially compact -> the gathered rows stay L2-resident) the positions are re-ordered by their
// 27-bit presence mask, so a 32-row MFMA block meets far fewer distinct (block, offset) combinations
// (measured on the synthetic rooms: zero-padded MFMA work 1.83x -> 1.31x of the real pairs).
constexpr int kMaskWindow = 16384;
// Sort key of a neighbourhood mask inside its window (tuning knob MASK_ORDER; the row order is free, only speed depends on it):
//   0  the mask itself (offset 26 most significant);
//   1  offsets by CLASS, corners most significant, then edges, faces, centre -- the rare offsets split the window first, so a
//      tile's rows agree on them and fewer (32-row block, offset) pairs are gathered for nothing;
//   2  the reverse (faces most significant);  3  popcount-major, then the mask.
__device__ inline uint32_t mask_sort_code(uint32_t m, int order) {
  if (order == 0) return m;
  if (order == 3) return ((uint32_t)__popc(m) << 27) | m;
  // offset k = (dx+1) + 3 (dy+1) + 9 (dz+1): class = number of non-zero components
  uint32_t corner = 0, edge = 0, face = 0, centre = (m >> 13) & 1u;
  int nc = 0, ne = 0, nf = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    const int cls = (dx != 0) + (dy != 0) + (dz != 0);
    const uint32_t b = (m >> k) & 1u;
    if (cls == 3) corner |= b << nc++;
    else if (cls == 2) edge |= b << ne++;
    else if (cls == 1) face |= b << nf++;
  }
  if (order == 1) return (corner << 19) | (edge << 7) | (face << 1) | centre;
  return (face << 21) | (edge << 9) | (corner << 1) | centre;
}
__global__ void k_mask_sort_keys(const uint32_t *pmask, int64_t n, int64_t n_pad, int window, int order, uint64_t *keys, int32_t *vals) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pad) return;
  keys[p] = p < n ? (((uint64_t)(p / window)) << 32) | mask_sort_code(pmask[p], order) : ~0ull;
  vals[p] = (int32_t)p;
}
// key bits the window sort has to look at: 32 bits of mask code + the window index.  A padding key is all ones; its low
// bits beat every real key as long as the largest real window index is not all ones too -- one more bit than it needs.
inline unsigned mask_sort_bits(int64_t n_pad, int window) {
  const uint64_t nw = (uint64_t)((n_pad + window - 1) / window);
  unsigned wb = 1;
  while ((1ull << wb) - 1 < nw) ++wb;
  return 32u + wb > 64u ? 64u : 32u + wb;
}
__global__ void k_permute_map3(const int32_t *nbr_tmp, const uint32_t *pmask, const int32_t *perm, const int32_t *order,
                               int64_t n, int64_t n_pad, int32_t *nbr, int32_t *out_row, uint32_t *mask64) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // grid covers n_pad exactly
  int32_t src = q < n ? perm[q] : -1;
  uint32_t pm = src >= 0 ? pmask[src] : 0u;
  out_row[q] = src >= 0 ? (order ? order[src] : src) : -1;
  uint32_t m = 0;
  for (int k = 0; k < 27; ++k) {
    nbr[(int64_t)k * n_pad + q] = src >= 0 ? nbr_tmp[(int64_t)k * n_pad + src] : -1;
    if (__ballot((pm >> k) & 1u)) m |= 1u << k;
  }
  if ((threadIdx.x & 63) == 0) mask64[q >> 6] = m;
}

// 2x2x2 stride-2, coarse-stationary view: nbr8[k][q] = fine row of child k of coarse row q
__global__ void k_build_map2_coarse(const uint64_t *fkeys, const int32_t *forder, const int32_t *cstart,
                                    int64_t n_c, int64_t nc_pad, int shift, int32_t *nbr8, uint32_t *mask64) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t child[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) child[k] = -1;
  if (q < n_c) {
    int32_t s = cstart[q], e = cstart[q + 1];
    for (int32_t p = s; p < e; ++p) {
      int k = (int)((fkeys[p] >> shift) & 7);
      int32_t row = forder ? forder[p] : p;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j == k) child[j] = row;
    }
  }
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    nbr8[(int64_t)k * nc_pad + q] = child[k];
    if (__ballot(child[k] >= 0)) m |= 1u << k;
  }
  if ((threadIdx.x & 63) == 0) mask64[q >> 6] = m;
}

// fine-grouped (degree-1) view: fine positions sorted by child index k, each group padded to 256
__global__ void k_child_keys(const uint64_t *fkeys, int64_t n, int shift, uint32_t *kk, int32_t *pp, int32_t *cnt) {
  __shared__ int32_t h[8];
  if (threadIdx.x < 8) h[threadIdx.x] = 0;
  __syncthreads();
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    uint32_t k = (uint32_t)((fkeys[p] >> shift) & 7);
    kk[p] = k;
    pp[p] = (int32_t)p;
    atomicAdd(&h[k], 1);
  }
  __syncthreads();
  if (threadIdx.x < 8 && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_group_offsets(const int32_t *cnt, int32_t *goff /*[9] padded starts*/, int32_t *gsrc /*[9] plain starts*/) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int32_t a = 0, b = 0;
    for (int k = 0; k < 8; ++k) {
      goff[k] = a; gsrc[k] = b;
      a += (cnt[k] + kPadRows - 1) / kPadRows * kPadRows;
      b += cnt[k];
    }
    goff[8] = a; gsrc[8] = b;
  }
}
__global__ void k_fill_i32(int32_t *p, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_build_map2_fine(const uint32_t *kk_sorted, const int32_t *pp_sorted, int64_t n, const int32_t *goff,
                                  const int32_t *gsrc, const int32_t *fine_cidx, const int32_t *forder,
                                  int32_t *g_nbr, int32_t *g_out, int32_t *tile_k) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int k = (int)kk_sorted[s];
  int32_t p = pp_sorted[s];
  int32_t slot = goff[k] + ((int32_t)s - gsrc[k]);
  g_nbr[slot] = fine_cidx[p];
  g_out[slot] = forder ? forder[p] : p;
  if (((s - gsrc[k]) & (kGroup - 1)) == 0) tile_k[slot >> 6] = k;
}

// export helpers (parity tests): count pairs of a view then write (k, in, out) triples
__global__ void k_view_count(View v, int32_t *count) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= v.n_pad) return;
  int32_t orow = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
  int c = 0;
  if (orow >= 0) {
    if (!v.nbr) c = 1;
    else
      for (int s = 0; s < v.KS; ++s) c += v.nbr[(int64_t)s * v.n_pad + p] >= 0;
  }
  if (c) atomicAdd(count, c);
}
__global__ void k_view_export(View v, int32_t *cursor, int32_t *ek, int32_t *ein, int32_t *eout) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= v.n_pad) return;
  int32_t orow = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
  if (orow < 0) return;
  for (int s = 0; s < v.KS; ++s) {
    int32_t i = v.nbr ? v.nbr[(int64_t)s * v.n_pad + p] : (int32_t)p;
    if (i < 0) continue;
    int k = v.tile_k ? v.tile_k[p >> 6] : (v.mirror ? v.K - 1 - s : s);
    int32_t at = atomicAdd(cursor, 1);
    ek[at] = k; ein[at] = i; eout[at] = orow;
  }
}

struct CoordMap {
  int ts = 1, log2ts = 0;
  int64_t n = 0, n_pad = 0;
  int32_t *coords = nullptr;  // [n,4]
  int32_t *order = nullptr;   // sorted position -> row; nullptr = identity
  uint64_t *skeys = nullptr;  // sorted Morton keys [n]
  uint64_t *hkeys = nullptr;  // hash (lazy)
  int32_t *hvals = nullptr;
  int64_t hcap = 0;
  int fine_key = -1, coarse_key = -1;
  int32_t *cstart = nullptr;     // [n+1] first fine sorted position of each row (maps made by stride2)
  int32_t *fine_cidx = nullptr;  // [n_fine] coarse row of each fine sorted position
};

}  // namespace lgs

using namespace lgs;

struct lgs_manager {
  int device = 0;
  hipMemPool_t pool = nullptr;
  // All map construction runs on the manager's OWN stream: the host-side row-count syncs then wait for map work
  // only (never for the compute backlog of the caller's stream), and the maps of step t+1 are built while step t's
  // backward is still running.  Consumers order themselves after `ev_ready` (lgs::kmap_wait).
  hipStream_t ms = nullptr;
  hipEvent_t ev_ready = nullptr, ev_in = nullptr;
  std::vector<hipStream_t> users;  // caller streams that consumed this manager's arrays (joined before freeing)
  hipStream_t last_stream = nullptr;
  std::vector<CoordMap> maps;
  std::vector<lgs_kmap *> kmaps;
  std::vector<void *> allocs;      // pool allocations (no arena yet, or the arena was too small)
  int *d_err = nullptr;
  // arena: one device block per manager, bump-allocated (see "arena" below)
  char *arena = nullptr;
  size_t arena_cap = 0, arena_used = 0, arena_peak = 0;
  struct Blk { size_t off, size; bool freed; };
  std::vector<Blk> blks;           // live arena blocks in allocation order (a stack: freed blocks on top are popped)
  size_t pool_live = 0, pool_peak = 0;
  int64_t precount[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // rows of the maps at tensor stride 2^(i+1), counted by the insert (-1: unknown)
};

namespace {

// the entry points run on the manager's device and leave the caller's current device as they found it
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// The maps of step t+1 reuse step t's memory: map arrays come from a PRIVATE stream-ordered pool per device whose
// release threshold is unlimited (the process-wide default pool is left alone).
hipMemPool_t g_pool[64] = {nullptr};
int ensure_pool(int device) {
  if (g_pool[device]) return 0;
  hipMemPoolProps props;
  memset(&props, 0, sizeof(props));
  props.allocType = hipMemAllocationTypePinned;
  props.handleTypes = hipMemHandleTypeNone;
  props.location.type = hipMemLocationTypeDevice;
  props.location.id = device;
  hipMemPool_t pool = nullptr;
  LGS_HIP(hipMemPoolCreate(&pool, &props));
  uint64_t thr = UINT64_MAX;
  LGS_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
  // blocks freed on a user stream (lgs_manager_destroy) are handed to the map stream only once that free has completed:
  // the allocator must not buy reuse with a stream dependency on the compute stream's tail
  int off = 0;
  (void)hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowInternalDependencies, &off);
  g_pool[device] = pool;
  return 0;
}
// ---- arena.  A manager makes ~150 device allocations per step; as stream-ordered pool calls each of them (and each
// free when the manager dies) is a marker packet in a HIP stream, and freeing a manager cost ~3 ms of stream time per
// step.  Instead every manager owns ONE block, bump-allocated on the host: allocation and release are pointer
// arithmetic; temporaries are released as a stack (a freed block is reclaimed once everything above it is freed too --
// all users of the arena run on the map stream, in order).  Blocks are recycled through a per-device FIFO: when a manager
// dies its block is stamped with one event per user stream and queued; a new manager takes the OLDEST queued block (its
// users finished a step ago, so the wait on its events is already satisfied) or, while fewer than two are queued,
// allocates a new one sized to the largest need seen so far.  Steady state: three blocks in rotation, no allocator
// call at all.  Whatever does not fit (first step, growing scenes) falls back to the private stream-ordered pool.
struct ArenaBlock { char *base; size_t cap; std::vector<hipEvent_t> ready; };
std::vector<ArenaBlock> g_arenas[64];   // FIFO of released blocks per device
size_t g_arena_need[64] = {0};          // largest (arena peak + pool peak) any manager of this device has needed

inline hipError_t pool_alloc(lgs_manager *m, void **q, size_t bytes, hipStream_t s) {
  return hipMallocFromPoolAsync(q, bytes, m->pool, s);
}
int raw_alloc(lgs_manager *m, void **p, size_t bytes, hipStream_t s) {
  bytes = (bytes + 255) / 256 * 256;
  if (m->arena && m->arena_used + bytes <= m->arena_cap) {
    *p = m->arena + m->arena_used;
    m->blks.push_back({m->arena_used, bytes, false});
    m->arena_used += bytes;
    if (m->arena_used > m->arena_peak) m->arena_peak = m->arena_used;
    return 0;
  }
  void *q = nullptr;
  LGS_HIP(pool_alloc(m, &q, bytes, s));
  m->allocs.push_back(q);
  m->pool_live += bytes;
  if (m->pool_live > m->pool_peak) m->pool_peak = m->pool_live;
  *p = q;
  return 0;
}
template <typename T>
int dalloc(lgs_manager *m, T **p, int64_t count, hipStream_t s) {
  void *q = nullptr;
  if (raw_alloc(m, &q, sizeof(T) * (size_t)(count > 0 ? count : 1), s)) return 1;
  *p = reinterpret_cast<T *>(q);
  return 0;
}
int dfree_now(lgs_manager *m, void *q, hipStream_t s) {  // temp buffer: release early
  char *c = reinterpret_cast<char *>(q);
  if (m->arena && c >= m->arena && c < m->arena + m->arena_cap) {
    const size_t off = (size_t)(c - m->arena);
    for (size_t i = m->blks.size(); i-- > 0;)
      if (m->blks[i].off == off) { m->blks[i].freed = true; break; }
    while (!m->blks.empty() && m->blks.back().freed) { m->arena_used = m->blks.back().off; m->blks.pop_back(); }
    return 0;
  }
  for (size_t i = 0; i < m->allocs.size(); ++i)
    if (m->allocs[i] == q) { m->allocs[i] = m->allocs.back(); m->allocs.pop_back(); break; }
  LGS_HIP(hipFreeAsync(q, s));
  return 0;
}
// take a recycled block (or a new one) for a fresh manager; the map stream waits for the block's previous users
int arena_acquire(lgs_manager *m) {
  const int dev = m->device;
  const size_t need = g_arena_need[dev];
  if (need == 0) return 0;                       // first manager of the device: measure through the pool
  std::vector<ArenaBlock> &q = g_arenas[dev];
  // never the most recently released block (its users -- the previous step's backward -- are still running)
  for (size_t i = 0; q.size() >= 2 && i + 1 < q.size(); ++i) {
    if (q[i].cap >= need) {
      ArenaBlock b = q[i];
      q.erase(q.begin() + (long)i);
      if (tune(T_ARENA_DBG)) {
        int pending = 0;
        for (hipEvent_t e : b.ready) pending += hipEventQuery(e) == hipSuccess ? 0 : 1;
        fprintf(stderr, "[arena] reuse block %zu of %zu (cap %zu MB, need %zu MB): %d of %zu events still pending\n", i, q.size() + 1, b.cap >> 20,
                need >> 20, pending, b.ready.size());
      }
      for (hipEvent_t e : b.ready) { (void)hipStreamWaitEvent(m->ms, e, 0); (void)hipEventDestroy(e); }
      m->arena = b.base; m->arena_cap = b.cap;
      return 0;
    }
    if (i == 0 && q.size() >= 4) {               // an undersized block at the head of a long queue: retire it
      for (hipEvent_t e : q[0].ready) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
      (void)hipFree(q[0].base);
      q.erase(q.begin());
      i = (size_t)-1;
    }
  }
  const size_t cap = (need + need / 4 + (2u << 20)) / (2u << 20) * (2u << 20);
  if (tune(T_ARENA_DBG)) fprintf(stderr, "[arena] hipMalloc %zu MB (need %zu MB, %zu queued)\n", cap >> 20, need >> 20, q.size());
  void *p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return 0; }   // no block: this manager uses the pool
  m->arena = reinterpret_cast<char *>(p); m->arena_cap = cap;
  return 0;
}
void arena_release(lgs_manager *m) {
  const size_t need = m->arena_peak + m->pool_peak;
  if (need > g_arena_need[m->device]) g_arena_need[m->device] = need;
  if (!m->arena) return;
  ArenaBlock b{m->arena, m->arena_cap, {}};
  std::vector<hipStream_t> streams = m->users;
  streams.push_back(m->ms);
  for (hipStream_t u : streams) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
      if (hipEventRecord(e, u) == hipSuccess) b.ready.push_back(e); else (void)hipEventDestroy(e);
    }
  }
  g_arenas[m->device].push_back(b);
  m->arena = nullptr;
}
void add_user(lgs_manager *m, hipStream_t caller) {
  for (hipStream_t u : m->users)
    if (u == caller) return;
  m->users.push_back(caller);
}
// inputs produced on the caller's stream must be complete before map work that reads them
int begin_from_caller(lgs_manager *m, hipStream_t caller) {
  LGS_HIP(hipEventRecord(m->ev_in, caller));
  LGS_HIP(hipStreamWaitEvent(m->ms, m->ev_in, 0));
  return 0;
}
// publish map work; if `caller` is given it is ordered after it (outputs written into caller-owned memory)
int publish(lgs_manager *m, hipStream_t caller, bool caller_waits) {
  LGS_HIP(hipEventRecord(m->ev_ready, m->ms));
  if (caller_waits) {
    LGS_HIP(hipStreamWaitEvent(caller, m->ev_ready, 0));
    add_user(m, caller);
  }
  return 0;
}
inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t > 0 ? (n + t - 1) / t : 1); }

int ensure_hash(lgs_manager *m, CoordMap &cm, hipStream_t s) {
  if (cm.hkeys) return 0;
  int64_t cap = 1024;
  while (cap < 2 * cm.n) cap <<= 1;
  cm.hcap = cap;
  if (dalloc(m, &cm.hkeys, cap, s)) return 1;
  if (dalloc(m, &cm.hvals, cap, s)) return 1;
  LGS_KLAUNCH(k_hash_fill, nblk(cap), 256, 0, s, cm.hkeys, cap);
  if (cm.n > 0)
    LGS_KLAUNCH(k_hash_insert, nblk(cm.n), 256, 0, s, cm.skeys, cm.order, cm.n, cm.hkeys, cm.hvals,
                       (uint64_t)(cap - 1));
  LGS_HIP(hipGetLastError());
  return 0;
}

int scan_incl(lgs_manager *m, const int32_t *in, int32_t *out, int64_t n, hipStream_t s) {
  size_t tb = 0;
  LGS_HIP(rocprim::inclusive_scan(nullptr, tb, in, out, (size_t)n, rocprim::plus<int32_t>(), s));
  void *tmp = nullptr;
  if (raw_alloc(m, &tmp, tb ? tb : 16, s)) return 1;
  LGS_HIP(rocprim::inclusive_scan(tmp, tb, in, out, (size_t)n, rocprim::plus<int32_t>(), s));
  if (dfree_now(m, tmp, s)) return 1;
  return 0;
}

}  // namespace

namespace lgs {
int kmap_wait(lgs_kmap *km, hipStream_t stream) {
  lgs_manager *m = km->mgr;
  LGS_HIP(hipStreamWaitEvent(stream, m->ev_ready, 0));
  add_user(m, stream);
  return 0;
}
}  // namespace lgs

extern "C" {

int lgs_abi_version(void) { return LGS_ABI_VERSION; }
const char *lgs_last_error(void) { return g_err.c_str(); }

int lgs_manager_create(int device, lgs_manager **out) {
  LGS_REQUIRE(out != nullptr, "lgs_manager_create: null out");
  LGS_REQUIRE(device >= 0 && device < 64, "lgs_manager_create: device index out of range");
  DeviceGuard guard(device);
  if (ensure_pool(device)) return 1;
  lgs_manager *m = new lgs_manager();
  m->device = device;
  m->pool = g_pool[device];
  // one map stream per device for the whole process (creating / destroying a stream per batch costs host time and
  // can block): managers of consecutive steps simply queue behind each other on it
  static hipStream_t g_map_stream[64] = {nullptr};
  if (!g_map_stream[device]) LGS_HIP(hipStreamCreateWithFlags(&g_map_stream[device], hipStreamNonBlocking));
  m->ms = g_map_stream[device];
  LGS_HIP(hipEventCreateWithFlags(&m->ev_ready, hipEventDisableTiming));
  LGS_HIP(hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming));
  arena_acquire(m);
  *out = m;
  return 0;
}

int lgs_manager_destroy(lgs_manager *m) {
  if (!m) return 0;
  DeviceGuard guard(m->device);
  // Every stream that read the maps must be done with them before the (stream-ordered) frees.  The frees are queued on
  // one of the USER streams (after joining the others and the manager's last map work), not on the map stream: making the
  // map stream wait for the users -- i.e. for the end of this step's backward -- would hold back the map construction of
  // the NEXT step, which is queued on that stream and is meant to run during this backward (measured: +3.5 ms per step).
  hipStream_t fs = m->users.empty() ? m->ms : m->users.back();
  if (!m->users.empty()) {
    for (hipStream_t u : m->users) {
      if (u == fs) continue;
      if (hipEventRecord(m->ev_in, u) == hipSuccess) (void)hipStreamWaitEvent(fs, m->ev_in, 0);
    }
    (void)hipStreamWaitEvent(fs, m->ev_ready, 0);   // the manager's own last map work (as recorded by its last publish)
  }
  for (void *p : m->allocs) (void)hipFreeAsync(p, fs);   // pool fallbacks only (none in the steady state)
  arena_release(m);                                      // the block goes back to the device's FIFO, stamped per user stream
  for (lgs_kmap *k : m->kmaps) delete k;
  (void)hipEventDestroy(m->ev_ready);
  (void)hipEventDestroy(m->ev_in);
  delete m;
  return 0;
}

int lgs_manager_insert(lgs_manager *m, const int32_t *coords, int64_t n, int64_t *unique_index, int64_t *inverse,
                       void *stream, int *key, int64_t *n_unique) {
  LGS_REQUIRE(m && key && n_unique, "lgs_manager_insert: null argument");
  LGS_REQUIRE(m->maps.empty(), "lgs_manager_insert: manager already holds a stride-1 map");
  LGS_REQUIRE(n >= 0 && n < (1ll << 31) - 1024, "lgs_manager_insert: row count out of range");
  hipStream_t caller = (hipStream_t)stream;
  hipStream_t s = m->ms;
  DeviceGuard guard(m->device);
  m->last_stream = caller;
  if (begin_from_caller(m, caller)) return 1;   // `coords` was produced on the caller's stream
  CoordMap cm;
  if (n == 0) {
    cm.n = 0; cm.n_pad = 0;
    m->maps.push_back(cm);
    *key = 0; *n_unique = 0;
    return 0;
  }
  if (!m->d_err) { if (dalloc(m, &m->d_err, 1, s)) return 1; }
  LGS_HIP(hipMemsetAsync(m->d_err, 0, sizeof(int), s));
  uint64_t *keys, *skeys; int32_t *vals, *svals, *head, *runid, *is_first, *urow;
  if (dalloc(m, &keys, n, s) || dalloc(m, &skeys, n, s) || dalloc(m, &vals, n, s) || dalloc(m, &svals, n, s) ||
      dalloc(m, &head, n, s) || dalloc(m, &runid, n, s) || dalloc(m, &is_first, n, s) || dalloc(m, &urow, n + 1, s))
    return 1;
  LGS_KLAUNCH(k_pack_keys, nblk(n), 256, 0, s, coords, n, keys, vals, m->d_err);
  {
    size_t tb = 0;
    LGS_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys, skeys, vals, svals, (size_t)n, 0, 64, s));
    void *tmp = nullptr;
    if (raw_alloc(m, &tmp, tb ? tb : 16, s)) return 1;
    LGS_HIP(rocprim::radix_sort_pairs(tmp, tb, keys, skeys, vals, svals, (size_t)n, 0, 64, s));
    if (dfree_now(m, tmp, s)) return 1;
  }
  LGS_KLAUNCH(k_heads, nblk(n), 256, 0, s, skeys, n, ~0ull, head);
  LGS_KLAUNCH(k_mark_first, nblk(n), 256, 0, s, svals, head, n, is_first);
  if (scan_incl(m, head, runid, n, s)) return 1;
  {  // exclusive scan of is_first = inclusive shifted: urow[0]=0, urow[i+1] = incl[i]
    LGS_HIP(hipMemsetAsync(urow, 0, sizeof(int32_t), s));
    if (scan_incl(m, is_first, urow + 1, n, s)) return 1;
  }
  int32_t *lvl_counts;
  if (dalloc(m, &lvl_counts, kPreLevels, s)) return 1;
  LGS_HIP(hipMemsetAsync(lvl_counts, 0, sizeof(int32_t) * kPreLevels, s));
  LGS_KLAUNCH(k_count_levels, (unsigned)((n + 256 * kCountPerThread - 1) / (256 * kCountPerThread)), 256, 0, s, skeys, n, lvl_counts);
  int32_t h_nu = 0; int h_err = 0; int32_t h_lvl[kPreLevels];
  LGS_HIP(hipMemcpyAsync(&h_nu, urow + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  LGS_HIP(hipMemcpyAsync(&h_err, m->d_err, sizeof(int), hipMemcpyDeviceToHost, s));
  LGS_HIP(hipMemcpyAsync(h_lvl, lvl_counts, sizeof(int32_t) * kPreLevels, hipMemcpyDeviceToHost, s));
  LGS_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < kPreLevels; ++i) m->precount[i] = h_lvl[i];
  LGS_REQUIRE(h_err == 0,
              "lgs_manager_insert: coordinate out of range (batch must be in [0,1024), |x|,|y|,|z| < 131008)");
  int64_t nu = h_nu;
  cm.n = nu; cm.n_pad = pad_rows(nu);
  if (dalloc(m, &cm.coords, nu * 4, s) || dalloc(m, &cm.order, nu, s) || dalloc(m, &cm.skeys, nu, s)) return 1;
  LGS_KLAUNCH(k_emit_unique, nblk(n), 256, 0, s, coords, is_first, urow, n, cm.coords, unique_index);
  LGS_KLAUNCH(k_emit_sorted, nblk(n), 256, 0, s, skeys, svals, head, runid, urow, n, cm.skeys, cm.order);
  if (inverse) LGS_KLAUNCH(k_emit_inverse, nblk(n), 256, 0, s, svals, runid, cm.order, n, inverse);
  LGS_HIP(hipGetLastError());
  if (dfree_now(m, keys, s) || dfree_now(m, skeys, s) || dfree_now(m, vals, s) || dfree_now(m, svals, s) ||
      dfree_now(m, head, s) || dfree_now(m, runid, s) || dfree_now(m, is_first, s) || dfree_now(m, urow, s) || dfree_now(m, lvl_counts, s))
    return 1;
  m->maps.push_back(cm);
  *key = 0; *n_unique = nu;
  return publish(m, caller, true);   // unique_index / inverse live in caller memory; `coords` may be reused after this
}

int lgs_manager_stride2(lgs_manager *m, int in_key, void *stream, int *out_key, int64_t *n_out) {
  LGS_REQUIRE(m && out_key && n_out, "lgs_manager_stride2: null argument");
  LGS_REQUIRE(in_key >= 0 && in_key < (int)m->maps.size(), "lgs_manager_stride2: bad key");
  hipStream_t s = m->ms;
  (void)stream;
  DeviceGuard guard(m->device);
  if (m->maps[in_key].coarse_key >= 0) {
    *out_key = m->maps[in_key].coarse_key; *n_out = m->maps[*out_key].n;
    return 0;
  }
  CoordMap f = m->maps[in_key];
  LGS_REQUIRE(f.log2ts < 12, "lgs_manager_stride2: tensor stride too large");
  CoordMap c;
  c.ts = f.ts * 2; c.log2ts = f.log2ts + 1; c.fine_key = in_key;
  int64_t n = f.n;
  if (n == 0) {
    m->maps.push_back(c);
    m->maps[in_key].coarse_key = (int)m->maps.size() - 1;
    *out_key = m->maps[in_key].coarse_key; *n_out = 0;
    return 0;
  }
  uint64_t keep = ~(7ull << (3 * f.log2ts));
  int32_t *head, *cincl;
  if (dalloc(m, &head, n, s) || dalloc(m, &cincl, n, s)) return 1;
  LGS_KLAUNCH(k_heads, nblk(n), 256, 0, s, f.skeys, n, keep, head);
  if (scan_incl(m, head, cincl, n, s)) return 1;
  int64_t nc;
  if (c.log2ts >= 1 && c.log2ts <= kPreLevels && m->precount[c.log2ts - 1] >= 0) {
    nc = m->precount[c.log2ts - 1];                 // counted by lgs_manager_insert: no host synchronisation here
  } else {
    int32_t h_nc = 0;
    LGS_HIP(hipMemcpyAsync(&h_nc, cincl + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    LGS_HIP(hipStreamSynchronize(s));
    nc = h_nc;
  }
  c.n = nc; c.n_pad = pad_rows(nc);
  if (dalloc(m, &c.coords, nc * 4, s) || dalloc(m, &c.skeys, nc, s) || dalloc(m, &c.cstart, nc + 1, s) ||
      dalloc(m, &c.fine_cidx, n, s))
    return 1;
  LGS_KLAUNCH(k_emit_coarse, nblk(n), 256, 0, s, f.skeys, head, cincl, n, keep, c.skeys, c.coords, c.cstart,
                     c.fine_cidx, nc, m->d_err);
  LGS_HIP(hipGetLastError());
  if (dfree_now(m, head, s) || dfree_now(m, cincl, s)) return 1;
  m->maps.push_back(c);
  int ck = (int)m->maps.size() - 1;
  m->maps[in_key].coarse_key = ck;
  *out_key = ck; *n_out = nc;
  return publish(m, nullptr, false);
}

int lgs_manager_check(lgs_manager *m, int *flags) {
  LGS_REQUIRE(m && flags, "lgs_manager_check: null argument");
  *flags = 0;
  if (!m->d_err) return 0;
  DeviceGuard guard(m->device);
  int h = 0;
  LGS_HIP(hipMemcpyAsync(&h, m->d_err, sizeof(int), hipMemcpyDeviceToHost, m->ms));
  LGS_HIP(hipStreamSynchronize(m->ms));
  *flags = h;
  return 0;
}

int lgs_manager_parent_of(lgs_manager *m, int key, int *fine_key) {
  LGS_REQUIRE(m && fine_key && key >= 0 && key < (int)m->maps.size(), "lgs_manager_parent_of: bad argument");
  *fine_key = m->maps[key].fine_key;
  return 0;
}

int lgs_manager_map_size(lgs_manager *m, int key, int64_t *n, int *tensor_stride) {
  LGS_REQUIRE(m && key >= 0 && key < (int)m->maps.size(), "lgs_manager_map_size: bad key");
  if (n) *n = m->maps[key].n;
  if (tensor_stride) *tensor_stride = m->maps[key].ts;
  return 0;
}

int lgs_manager_get_coords(lgs_manager *m, int key, int32_t *dst, void *stream) {
  LGS_REQUIRE(m && key >= 0 && key < (int)m->maps.size(), "lgs_manager_get_coords: bad key");
  const CoordMap &cm = m->maps[key];
  // `dst` was allocated on the caller's stream: a caching allocator may have handed out a block that kernels still
  // queued on that stream are using (e.g. a conv workspace released a moment ago), so the copy must be ordered after
  // the caller's pending work -- writing it early from the map stream corrupted a running conv's packed weights
  if (begin_from_caller(m, (hipStream_t)stream)) return 1;
  if (cm.n > 0)
    LGS_HIP(hipMemcpyAsync(dst, cm.coords, sizeof(int32_t) * 4 * (size_t)cm.n, hipMemcpyDeviceToDevice, m->ms));
  return publish(m, (hipStream_t)stream, true);
}

int lgs_manager_kernel_map(lgs_manager *m, int in_key, int out_key, int ks, void *stream, lgs_kmap **out) {
  LGS_REQUIRE(m && out, "lgs_manager_kernel_map: null argument");
  int nm = (int)m->maps.size();
  LGS_REQUIRE(in_key >= 0 && in_key < nm && out_key >= 0 && out_key < nm, "lgs_manager_kernel_map: bad key");
  for (lgs_kmap *k : m->kmaps)
    if (k->in_key == in_key && k->out_key == out_key && k->ks == ks) { *out = k; return 0; }
  hipStream_t s = m->ms;
  (void)stream;
  DeviceGuard guard(m->device);
  lgs_kmap *km = new lgs_kmap();
  km->mgr = m; km->in_key = in_key; km->out_key = out_key; km->ks = ks;
  CoordMap &ci = m->maps[in_key];
  if (ks == 1) {
    LGS_REQUIRE(in_key == out_key, "kernel_size 1 needs in_key == out_key");
    km->K = 1;
    View v; v.n_pad = ci.n_pad; v.n_out = ci.n; v.n_in = ci.n; v.KS = 1; v.K = 1;
    km->fwd = v; km->bwd = v;
  } else if (ks == 3) {
    LGS_REQUIRE(in_key == out_key, "kernel_size 3 is supported for stride 1 (in_key == out_key) only");
    km->K = 27;
    int32_t *nbr = nullptr, *orow = nullptr; uint32_t *mask = nullptr;
    if (ci.n > 0) {
      if (ensure_hash(m, ci, s)) return 1;
      int32_t *nbr_tmp, *vals, *perm; uint32_t *pmask; uint64_t *keys, *skeys2;
      if (dalloc(m, &nbr, 27 * ci.n_pad, s) || dalloc(m, &mask, ci.n_pad / kGroup, s) || dalloc(m, &orow, ci.n_pad, s) ||
          dalloc(m, &nbr_tmp, 27 * ci.n_pad, s) || dalloc(m, &pmask, ci.n_pad, s) || dalloc(m, &keys, ci.n_pad, s) ||
          dalloc(m, &skeys2, ci.n_pad, s) || dalloc(m, &vals, ci.n_pad, s) || dalloc(m, &perm, ci.n_pad, s))
        return 1;
      LGS_KLAUNCH(k_build_map3, (unsigned)(ci.n_pad / 256), 256, 0, s, ci.skeys, ci.n, ci.n_pad, ci.ts, ci.hkeys,
                         ci.hvals, (uint64_t)(ci.hcap - 1), nbr_tmp, pmask);
      const int window = (int)tune(T_MASK_WINDOW);   // tuning knob (default kMaskWindow)
      LGS_KLAUNCH(k_mask_sort_keys, (unsigned)(ci.n_pad / 256), 256, 0, s, pmask, ci.n, ci.n_pad, window, (int)tune(T_MASK_ORDER), keys, vals);
      {
        size_t tb = 0;
        const unsigned eb = mask_sort_bits(ci.n_pad, window);
        LGS_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys, skeys2, vals, perm, (size_t)ci.n_pad, 0, eb, s));
        void *tmp = nullptr;
        if (raw_alloc(m, &tmp, tb ? tb : 16, s)) return 1;
        LGS_HIP(rocprim::radix_sort_pairs(tmp, tb, keys, skeys2, vals, perm, (size_t)ci.n_pad, 0, eb, s));
        if (dfree_now(m, tmp, s)) return 1;
      }
      LGS_KLAUNCH(k_permute_map3, (unsigned)(ci.n_pad / 256), 256, 0, s, nbr_tmp, pmask, perm, ci.order, ci.n, ci.n_pad,
                         nbr, orow, mask);
      LGS_HIP(hipGetLastError());
      if (dfree_now(m, nbr_tmp, s) || dfree_now(m, pmask, s) || dfree_now(m, keys, s) || dfree_now(m, skeys2, s) ||
          dfree_now(m, vals, s) || dfree_now(m, perm, s))
        return 1;
    }
    View v; v.nbr = nbr; v.mask64 = mask; v.out_row = orow; v.n_pad = ci.n_pad; v.n_out = ci.n; v.n_in = ci.n;
    v.KS = 27; v.K = 27;
    km->fwd = v;
    km->bwd = v; km->bwd.mirror = 1;
  } else if (ks == 2) {
    CoordMap &co = m->maps[out_key];
    LGS_REQUIRE(co.fine_key == in_key, "kernel_size 2 needs out_key == stride2(in_key)");
    km->K = 8;
    int shift = 3 * ci.log2ts;
    View vf, vb;
    vf.KS = 8; vf.K = 8; vf.n_pad = co.n_pad; vf.n_out = co.n; vf.n_in = ci.n;
    vb.KS = 1; vb.K = 8; vb.n_out = ci.n; vb.n_in = co.n;
    if (ci.n > 0) {
      int32_t *nbr8; uint32_t *mask;
      if (dalloc(m, &nbr8, 8 * co.n_pad, s) || dalloc(m, &mask, co.n_pad / kGroup, s)) return 1;
      LGS_KLAUNCH(k_build_map2_coarse, (unsigned)(co.n_pad / 256), 256, 0, s, ci.skeys, ci.order, co.cstart, co.n,
                         co.n_pad, shift, nbr8, mask);
      vf.nbr = nbr8; vf.mask64 = mask;
      // grouped fine view
      int64_t n = ci.n, gp = pad_rows(n + 8 * kPadRows);
      uint32_t *kk, *kks; int32_t *pp, *pps, *cnt, *goff, *gsrc, *g_nbr, *g_out, *tile_k;
      if (dalloc(m, &kk, n, s) || dalloc(m, &kks, n, s) || dalloc(m, &pp, n, s) || dalloc(m, &pps, n, s) ||
          dalloc(m, &cnt, 8, s) || dalloc(m, &goff, 9, s) || dalloc(m, &gsrc, 9, s) || dalloc(m, &g_nbr, gp, s) ||
          dalloc(m, &g_out, gp, s) || dalloc(m, &tile_k, gp / kGroup, s))
        return 1;
      LGS_HIP(hipMemsetAsync(cnt, 0, 8 * sizeof(int32_t), s));
      LGS_KLAUNCH(k_child_keys, nblk(n), 256, 0, s, ci.skeys, n, shift, kk, pp, cnt);
      {
        size_t tb = 0;
        LGS_HIP(rocprim::radix_sort_pairs(nullptr, tb, kk, kks, pp, pps, (size_t)n, 0, 3, s));
        void *tmp = nullptr;
        if (raw_alloc(m, &tmp, tb ? tb : 16, s)) return 1;
        LGS_HIP(rocprim::radix_sort_pairs(tmp, tb, kk, kks, pp, pps, (size_t)n, 0, 3, s));
        if (dfree_now(m, tmp, s)) return 1;
      }
      LGS_KLAUNCH(k_group_offsets, 1, 64, 0, s, cnt, goff, gsrc);
      LGS_KLAUNCH(k_fill_i32, nblk(gp), 256, 0, s, g_nbr, gp, -1);
      LGS_KLAUNCH(k_fill_i32, nblk(gp), 256, 0, s, g_out, gp, -1);
      LGS_KLAUNCH(k_fill_i32, nblk(gp / kGroup), 256, 0, s, tile_k, gp / kGroup, -1);
      LGS_KLAUNCH(k_build_map2_fine, nblk(n), 256, 0, s, kks, pps, n, goff, gsrc, co.fine_cidx, ci.order, g_nbr,
                         g_out, tile_k);
      LGS_HIP(hipGetLastError());
      vb.nbr = g_nbr; vb.out_row = g_out; vb.tile_k = tile_k; vb.n_pad = gp;
      if (dfree_now(m, kk, s) || dfree_now(m, kks, s) || dfree_now(m, pp, s) || dfree_now(m, pps, s) ||
          dfree_now(m, cnt, s) || dfree_now(m, goff, s) || dfree_now(m, gsrc, s))
        return 1;
    }
    km->fwd = vf; km->bwd = vb;
  } else {
    delete km;
    LGS_REQUIRE(false, "unsupported kernel_size (the model family uses 1, 2 and 3 only)");
  }
  m->kmaps.push_back(km);
  *out = km;
  return publish(m, nullptr, false);
}

int lgs_kmap_export(lgs_kmap *km, int32_t *ek, int32_t *ein, int32_t *eout, void *stream, int64_t *mcount) {
  LGS_REQUIRE(km && mcount, "lgs_kmap_export: null argument");
  lgs_manager *m = km->mgr;
  hipStream_t caller = (hipStream_t)stream;
  hipStream_t s = m->ms;
  DeviceGuard guard(m->device);
  if (begin_from_caller(m, caller)) return 1;   // the output buffers were allocated on the caller's stream
  const View &v = km->fwd;
  int32_t *cnt;
  if (raw_alloc(m, (void **)&cnt, sizeof(int32_t), s)) return 1;
  LGS_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), s));
  if (v.n_pad > 0) {
    if (ek) LGS_KLAUNCH(k_view_export, nblk(v.n_pad), 256, 0, s, v, cnt, ek, ein, eout);
    else LGS_KLAUNCH(k_view_count, nblk(v.n_pad), 256, 0, s, v, cnt);
  }
  int32_t h = 0;
  LGS_HIP(hipMemcpyAsync(&h, cnt, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  LGS_HIP(hipStreamSynchronize(s));
  if (dfree_now(m, cnt, s)) return 1;
  *mcount = h;
  return publish(m, caller, true);
}

}  // extern "C"
