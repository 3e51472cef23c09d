// lgs_norm.hip -- fused BatchNorm (+ residual add) (+ ReLU) over sparse-tensor rows, gfx950.
//
// Replaces ME.MinkowskiBatchNorm (.bn = nn.BatchNorm1d on .F), MinkowskiReLU and `out += residual`
//   /root/reference/models/modules/common.py:17-19          get_norm()
//   /root/reference/models/modules/resnet_block.py:41-57     conv -> norm -> relu -> conv -> norm -> += -> relu
//   /root/reference/models/res16unet.py:196-270              conv -> bn -> relu chains
//
// Pure HBM-bound streaming work: features are [N, C] row-major with C in
// {32..512}; a thread owns a fixed group of 4 (fp32) or 8 (bf16) adjacent channels = one 16-byte
// access and walks rows, so every wave instruction is a fully coalesced 1 KiB access.  Statistics
// are reduced per block in LDS, then across blocks through a [blocks, 2C] fp32 scratch that a small fold
// kernel sums in a fixed order (double accumulation) -> deterministic, no float atomics.
#include "lgs_common.h"
#include <atomic>
#include <mutex>

namespace lgs {

__device__ inline float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ inline uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int W = 4;
  __device__ static void load(const float *p, float (&v)[4]) {
    float4 x = *reinterpret_cast<const float4 *>(p);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  __device__ static void store(float *p, const float (&v)[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ static void store_nt(float *p, const float (&v)[4]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 x = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(x, reinterpret_cast<f4 *>(p));
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int W = 8;
  __device__ static void load(const bf16_t *p, float (&v)[8]) {
    uint4 x = *reinterpret_cast<const uint4 *>(p);
    uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f((uint16_t)(w[i] & 0xffff)); v[2 * i + 1] = bf2f((uint16_t)(w[i] >> 16)); }
  }
  __device__ static void store(bf16_t *p, const float (&v)[8]) {
    uint4 x;
    x.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    x.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    x.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
    x.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = x;
  }
  __device__ static void store_nt(bf16_t *p, const float (&v)[8]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 x;
    x.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    x.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    x.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
    x.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    __builtin_nontemporal_store(x, reinterpret_cast<u4 *>(p));
  }
};

constexpr int kNT = 256;

// Column reduction of up to two per-element quantities.  MODE 0: (x, x*x)  [forward statistics]
// MODE 1: (dy', dy' * xhat) where dy' = dy masked by (y > 0) when relu  [backward reductions]
// Each block handles a contiguous slab of rows; thread layout: cg = tid % G channel groups, rl = tid / G.
// out: scratch[blocks][2][C]
// relu: 0 = none, 1 = mask from the saved output y (needed when a residual was added), 2 = mask recomputed from x
// as (xhat * gamma + beta > 0) -- one tensor read fewer.
template <typename T, int MODE>
__global__ __launch_bounds__(kNT) void k_colreduce(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy,
                                                   const float *__restrict__ stats, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, int64_t n, int c, int relu,
                                                   int64_t rows_per_block, float *__restrict__ scratch, int64_t dy_ld, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  const int G = c / W;              // channel groups per row
  const int RL = kNT / G;           // rows in flight per block iteration (G <= 256)
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  const bool active = rl < RL;
  float s0[W], s1[W];
#pragma unroll
  for (int i = 0; i < W; ++i) s0[i] = s1[i] = 0.f;
  float mean[W], istd[W], gm[W], bt[W];
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      mean[i] = stats[cg * W + i]; istd[i] = stats[c + cg * W + i];
      gm[i] = relu == 2 ? gamma[cg * W + i] : 0.f; bt[i] = relu == 2 ? beta[cg * W + i] : 0.f;
    }
  } else {
    // forward statistics are accumulated about a per-channel pivot (row 0, the same for every block) so that
    // var = E[(x-k)^2] - E[x-k]^2 does not cancel catastrophically when |mean| >> std
    if (n > 0) Vec<T>::load(x + cg * W, mean);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n);
  if (active) {
    int64_t r = r0 + rl;
    // U rows per thread are in flight before the first is consumed (a streaming reduction needs ~50 KB of loads in
    // flight per CU to cover the HBM latency); sums are then taken in row order, i.e. exactly as a 1-row loop would
    constexpr int U = MODE == 0 ? 4 : 2;
    auto accumulate = [&](const float (&xv)[W], float (&gv)[W], const float (&yv)[W]) __attribute__((always_inline)) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < W; ++i) { const float d = xv[i] - mean[i]; s0[i] += d; s1[i] += d * d; }
      } else {
        if (relu == 1) {
#pragma unroll
          for (int i = 0; i < W; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
        } else if (relu == 2) {
#pragma unroll
          for (int i = 0; i < W; ++i) gv[i] = ((xv[i] - mean[i]) * (istd[i] * gm[i]) + bt[i]) > 0.f ? gv[i] : 0.f;  // same expression as k_bn_apply
        }
#pragma unroll
        for (int i = 0; i < W; ++i) { s0[i] += gv[i]; s1[i] += gv[i] * (xv[i] - mean[i]) * istd[i]; }
      }
    };
    for (; r + (U - 1) * RL < r1; r += U * RL) {
      float xv[U][W], gv[U][W], yv[U][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t o = (r + u * RL) * c + cg * W;
        Vec<T>::load(x + o, xv[u]);
        if (MODE == 1) {
          Vec<T>::load(dy + (r + u * RL) * dy_ld + cg * W, gv[u]);
          if (relu == 1) Vec<T>::load(y + (r + u * RL) * y_ld + cg * W, yv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(xv[u], gv[u], yv[u]);
    }
    for (; r < r1; r += RL) {
      float xv[W], gv[W], yv[W];
      const int64_t o = r * c + cg * W;
      Vec<T>::load(x + o, xv);
      if (MODE == 1) {
        Vec<T>::load(dy + r * dy_ld + cg * W, gv);
        if (relu == 1) Vec<T>::load(y + r * y_ld + cg * W, yv);
      }
      accumulate(xv, gv, yv);
    }
  }
  // block reduction over rl through LDS
  __shared__ float red[2][kNT][8];
#pragma unroll
  for (int i = 0; i < W; ++i) { red[0][threadIdx.x][i] = s0[i]; red[1][threadIdx.x][i] = s1[i]; }
  __syncthreads();
  if (rl == 0) {
    for (int j = 1; j < RL; ++j) {
#pragma unroll
      for (int i = 0; i < W; ++i) { s0[i] += red[0][j * G + cg][i]; s1[i] += red[1][j * G + cg][i]; }
    }
    float *dst = scratch + (int64_t)blockIdx.x * 2 * c;
#pragma unroll
    for (int i = 0; i < W; ++i) { dst[cg * W + i] = s0[i]; dst[c + cg * W + i] = s1[i]; }
  }
}

// fold the per-block partials (fixed order, deterministic) and finish: forward -> mean / invstd / running stats,
// backward -> dbeta = sum dy', dgamma = sum dy' xhat.  Block = 16 channels x 16 partial-slices: a thread sums every
// 16th partial with eight independent loads in flight (the fold is a pure latency chain: 4 slices of 128 dependent
// steps took 13 us per BatchNorm, twice per layer), slices are combined through LDS in slice order.
constexpr int kFoldCh = 16, kFoldSl = 16;
__device__ inline void fold_sums(const float *__restrict__ scratch, int nblocks, int c, int ch, int part, double &s, double &ss,
                                 double (*red)[2][kFoldCh]) {
  s = 0.0; ss = 0.0;
  if (ch < c) {
#pragma unroll 8
    for (int b = part; b < nblocks; b += kFoldSl) { s += scratch[(int64_t)b * 2 * c + ch]; ss += scratch[(int64_t)b * 2 * c + c + ch]; }
  }
  red[part][0][threadIdx.x % kFoldCh] = s;
  red[part][1][threadIdx.x % kFoldCh] = ss;
  __syncthreads();
  if (part == 0) {
    for (int q = 1; q < kFoldSl; ++q) { s += red[q][0][threadIdx.x % kFoldCh]; ss += red[q][1][threadIdx.x % kFoldCh]; }
  }
}
// statistics that arrive as per-tile partial rows from the conv epilogue (BnEpi): fold groups of rows into <= 128 rows
// of the same [row][2][C] layout, fixed order (deterministic)
__global__ __launch_bounds__(256) void k_partial_reduce(const float *__restrict__ part, int rows, int c2, int rows_per_block,
                                                        float *__restrict__ out) {
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  for (int col = threadIdx.x; col < c2; col += blockDim.x) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
      a0 += part[(int64_t)(r + 0) * c2 + col]; a1 += part[(int64_t)(r + 1) * c2 + col];
      a2 += part[(int64_t)(r + 2) * c2 + col]; a3 += part[(int64_t)(r + 3) * c2 + col];
    }
    for (; r < r1; ++r) a0 += part[(int64_t)r * c2 + col];
    out[(int64_t)blockIdx.x * c2 + col] = (a0 + a1) + (a2 + a3);
  }
}

// pivot_mode 0: the per-channel pivot is row 0 of x (k_colreduce<0>); 1: pivot_ptr[ch] (or 0 if NULL) -- conv-epilogue partials
template <typename T>
__global__ __launch_bounds__(256) void k_fold_fwd(const float *__restrict__ scratch, const T *__restrict__ x, int nblocks, int c,
                                                  int64_t n, float eps, float momentum, float *__restrict__ running_mean,
                                                  float *__restrict__ running_var, long long *__restrict__ nbt,
                                                  float *__restrict__ stats, int pivot_mode, const float *__restrict__ pivot_ptr) {
  __shared__ double red[kFoldSl][2][kFoldCh];
  const int ch = blockIdx.x * kFoldCh + (threadIdx.x % kFoldCh), part = threadIdx.x / kFoldCh;
  double s, ss;
  fold_sums(scratch, nblocks, c, ch, part, s, ss, red);
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;   // nn.BatchNorm1d.num_batches_tracked
  if (part != 0 || ch >= c) return;
  const double pivot = pivot_mode ? (pivot_ptr ? (double)pivot_ptr[ch] : 0.0) : (n > 0 ? (double)ld_elem(x + ch) : 0.0);
  double dm = n > 0 ? s / (double)n : 0.0;
  double var = n > 0 ? ss / (double)n - dm * dm : 0.0;
  if (var < 0.0) var = 0.0;
  const double mean = pivot + dm;
  stats[ch] = (float)mean;
  stats[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
  }
}
// split API (SyncBN): local mean and M2 = sum (x - mean)^2, to be combined across ranks with Chan's formula
template <typename T>
__global__ __launch_bounds__(256) void k_fold_stats(const float *__restrict__ scratch, const T *__restrict__ x, int nblocks, int c,
                                                    int64_t n, float *__restrict__ mean_m2, int pivot_mode,
                                                    const float *__restrict__ pivot_ptr) {
  __shared__ double red[kFoldSl][2][kFoldCh];
  const int ch = blockIdx.x * kFoldCh + (threadIdx.x % kFoldCh), part = threadIdx.x / kFoldCh;
  double s, ss;
  fold_sums(scratch, nblocks, c, ch, part, s, ss, red);
  if (part != 0 || ch >= c) return;
  const double pivot = pivot_mode ? (pivot_ptr ? (double)pivot_ptr[ch] : 0.0) : (n > 0 ? (double)ld_elem(x + ch) : 0.0);
  const double dm = n > 0 ? s / (double)n : 0.0;
  double m2 = ss - (double)n * dm * dm;
  if (m2 < 0.0) m2 = 0.0;
  mean_m2[ch] = (float)(pivot + dm);
  mean_m2[c + ch] = (float)m2;
  if (ch == 0) mean_m2[2 * c] = (float)n;   // [mean | M2 | count]: one packed all-gather per layer
}

__global__ __launch_bounds__(256) void k_fold_bwd(const float *__restrict__ scratch, int nblocks, int c, float *__restrict__ dgamma,
                                                  float *__restrict__ dbeta, float *__restrict__ sums) {
  __shared__ double red[kFoldSl][2][kFoldCh];
  const int ch = blockIdx.x * kFoldCh + (threadIdx.x % kFoldCh), part = threadIdx.x / kFoldCh;
  double s, ss;
  fold_sums(scratch, nblocks, c, ch, part, s, ss, red);
  if (part != 0 || ch >= c) return;
  dbeta[ch] = (float)s;
  dgamma[ch] = (float)ss;
  sums[ch] = (float)s;
  sums[c + ch] = (float)ss;
}

// SyncBN: combine the per-rank [mean | M2 | count] records (Chan's parallel formula, double) into the global
// mean / invstd, update the running statistics and num_batches_tracked, and leave 1/N for the backward pass
__global__ void k_sync_combine(const float *__restrict__ all_stats, int world, int c, float eps, float momentum,
                               float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt,
                               float *__restrict__ stats, float *__restrict__ inv_n_out) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  const int rec = 2 * c + 1;
  double n = 0.0;
  for (int r = 0; r < world; ++r) n += (double)all_stats[(int64_t)r * rec + 2 * c];
  if (ch == 0) {
    if (inv_n_out) *inv_n_out = n > 0.0 ? (float)(1.0 / n) : 0.f;
    if (nbt) *nbt += 1;
  }
  if (ch >= c) return;
  double mean = 0.0;
  for (int r = 0; r < world; ++r) mean += (double)all_stats[(int64_t)r * rec + 2 * c] * (double)all_stats[(int64_t)r * rec + ch];
  mean = n > 0.0 ? mean / n : 0.0;
  double m2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double d = (double)all_stats[(int64_t)r * rec + ch] - mean;
    m2 += (double)all_stats[(int64_t)r * rec + c + ch] + (double)all_stats[(int64_t)r * rec + 2 * c] * d * d;
  }
  const double var = n > 0.0 ? m2 / n : 0.0;
  stats[ch] = (float)mean;
  stats[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = n > 1.0 ? m2 / (n - 1.0) : var;
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
  }
}

// y = relu?( (x - mean) * invstd * gamma + beta (+ residual) )
// A thread owns ONE channel group (its per-channel constants live in registers) and walks rows: with
// G = C / W groups, thread t handles group t % G of rows t / G, t / G + R, ...  (R = rows per sweep of the grid).
// Two independent 16-byte loads are in flight per operand per thread.
template <typename T>
__global__ __launch_bounds__(kNT) void k_bn_apply(const T *__restrict__ x, const T *__restrict__ res, int64_t n, int c,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  const float *__restrict__ stats, int relu, T *__restrict__ y, int64_t y_ld) {
  // y_ld = row stride of the output (elements): > c when y is a column slice of a wider buffer (zero-copy ME.cat)
  constexpr int W = Vec<T>::W;
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  if (rl >= RL) return;
  float sc[W], mean[W], bt[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int ch = cg * W + k;
    sc[k] = stats[c + ch] * gamma[ch];
    mean[k] = stats[ch];
    bt[k] = beta[ch];
  }
  const int64_t stride = (int64_t)gridDim.x * RL;
  for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < n; r += 2 * stride) {
    const int64_t r2 = r + stride;
    const bool two = r2 < n;
    float xa[W], xb[W], ra[W], rb[W];
    Vec<T>::load(x + r * c + cg * W, xa);
    if (two) Vec<T>::load(x + r2 * c + cg * W, xb);
    if (res) { Vec<T>::load(res + r * c + cg * W, ra); if (two) Vec<T>::load(res + r2 * c + cg * W, rb); }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      float o = (xa[k] - mean[k]) * sc[k] + bt[k];   // keep this exact expression: the backward recomputes the ReLU mask from it
      if (res) o += ra[k];
      xa[k] = (relu && o < 0.f) ? 0.f : o;
    }
    Vec<T>::store_nt(y + r * y_ld + cg * W, xa);
    if (two) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        float o = (xb[k] - mean[k]) * sc[k] + bt[k];
        if (res) o += rb[k];
        xb[k] = (relu && o < 0.f) ? 0.f : o;
      }
      Vec<T>::store_nt(y + r2 * y_ld + cg * W, xb);
    }
  }
}

// dx = gamma*invstd * (dy' - mean(dy') - xhat * mean(dy' xhat));  dres = dy'
template <typename T>
__global__ __launch_bounds__(kNT) void k_bn_bwd_apply(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy,
                                                      int64_t n, int c, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta,
                                                      const float *__restrict__ stats, const float *__restrict__ sums,
                                                      float inv_n, int relu, T *__restrict__ dx, T *__restrict__ dres,
                                                      int64_t dy_ld, const float *__restrict__ inv_n_dev, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  if (rl >= RL) return;
  float mean[W], istd[W], sc[W], bt[W], gi[W], m1[W], m2[W];
  if (inv_n_dev) inv_n = *inv_n_dev;   // SyncBN: 1 / global row count lives on the device (no host sync)
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int ch = cg * W + k;
    mean[k] = stats[ch]; istd[k] = stats[c + ch];
    sc[k] = istd[k] * (relu == 2 ? gamma[ch] : 0.f);
    bt[k] = relu == 2 ? beta[ch] : 0.f;
    gi[k] = gamma[ch] * istd[k];
    m1[k] = sums[ch] * inv_n; m2[k] = sums[c + ch] * inv_n;
  }
  const int64_t stride = (int64_t)gridDim.x * RL;
  for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < n; r += stride) {
    const int64_t o = r * c + cg * W;
    float xv[W], gv[W];
    Vec<T>::load(x + o, xv);
    Vec<T>::load(dy + r * dy_ld + cg * W, gv);
    if (relu == 1) {
      float yv[W];
      Vec<T>::load(y + r * y_ld + cg * W, yv);
#pragma unroll
      for (int k = 0; k < W; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    } else if (relu == 2) {
#pragma unroll
      for (int k = 0; k < W; ++k) gv[k] = ((xv[k] - mean[k]) * sc[k] + bt[k]) > 0.f ? gv[k] : 0.f;  // same expression as k_bn_apply
    }
    if (dres) Vec<T>::store_nt(dres + o, gv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float xh = (xv[k] - mean[k]) * istd[k];
      xv[k] = gi[k] * (gv[k] - m1[k] - xh * m2[k]);
    }
    Vec<T>::store_nt(dx + o, xv);
  }
}

inline int reduce_blocks(int64_t n, int64_t *rows_per_block) {
  // >= 128 rows per block, <= 512 blocks: coarse levels (a few thousand rows) still get tens of blocks -- with 1024
  // rows per block their reductions were 5-block, 13 us latency chains
  int64_t nb = (n + 127) / 128;
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  *rows_per_block = (n + nb - 1) / nb;
  if (*rows_per_block < 1) *rows_per_block = 1;
  return (int)((n + *rows_per_block - 1) / *rows_per_block > 0 ? (n + *rows_per_block - 1) / *rows_per_block : 1);
}

// ------------------------------------------------------------------------------------------------ one launch per direction
// A BatchNorm direction is three dependent steps (column sums over all rows -> fold + finish the statistics -> elementwise
// apply): three launches per direction made 372 of the compute stream's 596 launches per training step, most of them on
// coarse levels where a launch is a few microseconds of work.  The fused kernels run the three steps in ONE launch of <= 256
// co-resident workgroups separated by two grid-wide barriers (arrive = agent-scope add on a counter, wait = spin on
// it); a workgroup applies to the SAME slab of rows it reduced, walking it backwards, so the rows it read last in step 1 are
// the first it needs in step 3 (L2 / Infinity Cache hits instead of a second HBM pass on the large levels).
// Co-residency: 256 workgroups x 256 threads at <= 128 VGPRs is a quarter of what the chip holds (256 CUs x 4), so such kernels
// from different streams / processes can spin at the same time without starving each other's unscheduled workgroups.  The cap
// is 256 and not 512 because the weight-gradient kernel on the side stream owns whole CUs (all registers, all LDS) for up to a
// millisecond: every workgroup of a barrier kernel has to find a free CU before ANY of them gets past the first barrier
// (measured in the 8-scene step, layers <= 24 MB fused: 30.9 ms with 512 workgroups, 29.8 ms with 256, 30.0 ms unfused).
constexpr int kFusedMaxBlocks = 256;

// Values that cross workgroups inside a fused kernel (scratch rows, statistics) are written and read with agent-scope
// accesses (sc1: coherent across the eight XCDs' L2s) instead of fencing: a release / acquire fence at agent scope writes back
// and invalidates the whole L2 of the XCD, and 512 workgroups doing that twice cost 0.24 ms per launch.
__device__ inline void st_coh(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_coh(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void grid_arrive_wait(unsigned *ctr, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's coherent stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
// the last workgroup to leave puts the counter back to zero for the slot's next user
__device__ inline void grid_leave(unsigned *ctr, unsigned total) {
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == total - 1) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// fold_sums over rows other workgroups of this launch wrote
__device__ inline void fold_sums_coh(const float *scratch, int nblocks, int c, int ch, int part, double &s, double &ss,
                                     double (*red)[2][kFoldCh]) {
  s = 0.0; ss = 0.0;
  if (ch < c) {
#pragma unroll 8
    for (int b = part; b < nblocks; b += kFoldSl) { s += ld_coh(scratch + (int64_t)b * 2 * c + ch); ss += ld_coh(scratch + (int64_t)b * 2 * c + c + ch); }
  }
  red[part][0][threadIdx.x % kFoldCh] = s;
  red[part][1][threadIdx.x % kFoldCh] = ss;
  __syncthreads();
  if (part == 0) {
    for (int q = 1; q < kFoldSl; ++q) { s += red[q][0][threadIdx.x % kFoldCh]; ss += red[q][1][threadIdx.x % kFoldCh]; }
  }
}

// column sums of one slab of rows [r0, r1) -> dst[2][C]; the body of k_colreduce (same summation order)
template <typename T, int MODE>
__device__ inline void colreduce_slab(const T *x, const T *y, const T *dy, const float *stats, const float *gamma, const float *beta,
                                      int64_t n, int c, int relu, int64_t r0, int64_t r1, float *dst, int64_t dy_ld, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  const bool active = rl < RL;
  float s0[W], s1[W];
#pragma unroll
  for (int i = 0; i < W; ++i) s0[i] = s1[i] = 0.f;
  float mean[W], istd[W], gm[W], bt[W];
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      mean[i] = stats[cg * W + i]; istd[i] = stats[c + cg * W + i];
      gm[i] = relu == 2 ? gamma[cg * W + i] : 0.f; bt[i] = relu == 2 ? beta[cg * W + i] : 0.f;
    }
  } else {
    if (n > 0) Vec<T>::load(x + cg * W, mean);   // pivot = row 0, as in k_colreduce<T, 0>
  }
  if (active) {
    int64_t r = r0 + rl;
    constexpr int U = MODE == 0 ? 4 : 2;
    auto accumulate = [&](const float (&xv)[W], float (&gv)[W], const float (&yv)[W]) __attribute__((always_inline)) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < W; ++i) { const float d = xv[i] - mean[i]; s0[i] += d; s1[i] += d * d; }
      } else {
        if (relu == 1) {
#pragma unroll
          for (int i = 0; i < W; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
        } else if (relu == 2) {
#pragma unroll
          for (int i = 0; i < W; ++i) gv[i] = ((xv[i] - mean[i]) * (istd[i] * gm[i]) + bt[i]) > 0.f ? gv[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) { s0[i] += gv[i]; s1[i] += gv[i] * (xv[i] - mean[i]) * istd[i]; }
      }
    };
    for (; r + (U - 1) * RL < r1; r += U * RL) {
      float xv[U][W], gv[U][W], yv[U][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Vec<T>::load(x + (r + u * RL) * c + cg * W, xv[u]);
        if (MODE == 1) {
          Vec<T>::load(dy + (r + u * RL) * dy_ld + cg * W, gv[u]);
          if (relu == 1) Vec<T>::load(y + (r + u * RL) * y_ld + cg * W, yv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(xv[u], gv[u], yv[u]);
    }
    for (; r < r1; r += RL) {
      float xv[W], gv[W], yv[W];
      Vec<T>::load(x + r * c + cg * W, xv);
      if (MODE == 1) {
        Vec<T>::load(dy + r * dy_ld + cg * W, gv);
        if (relu == 1) Vec<T>::load(y + r * y_ld + cg * W, yv);
      }
      accumulate(xv, gv, yv);
    }
  }
  __shared__ float red[2][kNT][8];
#pragma unroll
  for (int i = 0; i < W; ++i) { red[0][threadIdx.x][i] = s0[i]; red[1][threadIdx.x][i] = s1[i]; }
  __syncthreads();
  if (rl == 0) {
    for (int j = 1; j < RL; ++j) {
#pragma unroll
      for (int i = 0; i < W; ++i) { s0[i] += red[0][j * G + cg][i]; s1[i] += red[1][j * G + cg][i]; }
    }
#pragma unroll
    for (int i = 0; i < W; ++i) { st_coh(dst + cg * W + i, s0[i]); st_coh(dst + c + cg * W + i, s1[i]); }
  }
}

template <typename T>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(4))) void k_bn_fwd_fused(const T *x, const T *res, int64_t n, int c, const float *gamma, const float *beta,
                                                      float eps, float momentum, float *running_mean, float *running_var,
                                                      long long *nbt, float *stats, int relu, T *y, int64_t y_ld, float *scratch,
                                                      const float *partials, int partial_rows, int partial_rpb, int nfoldrows,
                                                      const float *pivot_ptr, int64_t rows_per_block, unsigned *ctr) {
  constexpr int W = Vec<T>::W;
  const unsigned nwg = gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n);
  // ---- step 1: per-workgroup column sums (or: fold the conv epilogue's partial rows into <= 128 rows)
  if (partials) {
    const int c2 = 2 * c;
    for (int pb = blockIdx.x; pb < nfoldrows; pb += nwg) {
      const int p0 = pb * partial_rpb, p1 = min(p0 + partial_rpb, partial_rows);
      for (int col = threadIdx.x; col < c2; col += kNT) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = p0;
        for (; r + 4 <= p1; r += 4) {
          a0 += partials[(int64_t)(r + 0) * c2 + col]; a1 += partials[(int64_t)(r + 1) * c2 + col];
          a2 += partials[(int64_t)(r + 2) * c2 + col]; a3 += partials[(int64_t)(r + 3) * c2 + col];
        }
        for (; r < p1; ++r) a0 += partials[(int64_t)r * c2 + col];
        st_coh(scratch + (int64_t)pb * c2 + col, (a0 + a1) + (a2 + a3));
      }
    }
  } else {
    colreduce_slab<T, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, n, c, 0, r0, r1, scratch + (int64_t)blockIdx.x * 2 * c, c, c);
  }
  grid_arrive_wait(ctr, nwg);
  // ---- step 2: fold + finish the statistics, 16 channels per workgroup
  {
    __shared__ double red[kFoldSl][2][kFoldCh];
    const int nfb = (c + kFoldCh - 1) / kFoldCh;
    for (int fb = blockIdx.x; fb < nfb; fb += nwg) {
      const int ch = fb * kFoldCh + (threadIdx.x % kFoldCh), part = threadIdx.x / kFoldCh;
      double s, ss;
      fold_sums_coh(scratch, nfoldrows, c, ch, part, s, ss, red);
      if (part == 0 && ch < c) {
        const double pivot = partials ? (pivot_ptr ? (double)pivot_ptr[ch] : 0.0) : (n > 0 ? (double)ld_elem(x + ch) : 0.0);
        const double dm = n > 0 ? s / (double)n : 0.0;
        double var = n > 0 ? ss / (double)n - dm * dm : 0.0;
        if (var < 0.0) var = 0.0;
        const double mean = pivot + dm;
        st_coh(stats + ch, (float)mean);
        st_coh(stats + c + ch, (float)(1.0 / sqrt(var + (double)eps)));
        if (running_mean) {
          const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
          running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
          running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
        }
      }
      __syncthreads();
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  }
  grid_arrive_wait(ctr, 2 * nwg);
  grid_leave(ctr, 3 * nwg);
  // ---- step 3: y = relu?((x - mean) * invstd * gamma + beta (+ residual)) over this workgroup's slab, last rows first
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  if (rl >= RL) return;
  float sc[W], mean[W], bt[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int ch = cg * W + k;
    sc[k] = ld_coh(stats + c + ch) * gamma[ch];
    mean[k] = ld_coh(stats + ch);
    bt[k] = beta[ch];
  }
  constexpr int U = W == 8 ? 2 : 4;   // rows in flight per thread (registers: <= 128 VGPRs, see above)
  for (int64_t r = r1 - 1 - rl; r >= r0; r -= (int64_t)U * RL) {
    float xv[U][W], rv[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ru = r - (int64_t)u * RL;
      if (ru >= r0) {
        Vec<T>::load(x + ru * c + cg * W, xv[u]);
        if (res) Vec<T>::load(res + ru * c + cg * W, rv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ru = r - (int64_t)u * RL;
      if (ru >= r0) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
          float o = (xv[u][k] - mean[k]) * sc[k] + bt[k];   // the exact expression of k_bn_apply (the backward recomputes the mask from it)
          if (res) o += rv[u][k];
          xv[u][k] = (relu && o < 0.f) ? 0.f : o;
        }
        Vec<T>::store(y + ru * y_ld + cg * W, xv[u]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(4))) void k_bn_bwd_fused(const T *x, const T *y, const T *dy, int64_t n, int c, const float *gamma,
                                                      const float *beta, const float *stats, int relu, T *dx, T *dres, float *dgamma,
                                                      float *dbeta, float *scratch, float *sums, int64_t dy_ld, int64_t y_ld,
                                                      int64_t rows_per_block, float inv_n, unsigned *ctr) {
  constexpr int W = Vec<T>::W;
  const unsigned nwg = gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n);
  colreduce_slab<T, 1>(x, y, dy, stats, gamma, beta, n, c, relu, r0, r1, scratch + (int64_t)blockIdx.x * 2 * c, dy_ld, y_ld);
  grid_arrive_wait(ctr, nwg);
  {
    __shared__ double red[kFoldSl][2][kFoldCh];
    const int nfb = (c + kFoldCh - 1) / kFoldCh;
    for (int fb = blockIdx.x; fb < nfb; fb += nwg) {
      const int ch = fb * kFoldCh + (threadIdx.x % kFoldCh), part = threadIdx.x / kFoldCh;
      double s, ss;
      fold_sums_coh(scratch, (int)nwg, c, ch, part, s, ss, red);
      if (part == 0 && ch < c) {
        dbeta[ch] = (float)s; dgamma[ch] = (float)ss;
        st_coh(sums + ch, (float)s); st_coh(sums + c + ch, (float)ss);
      }
      __syncthreads();
    }
  }
  grid_arrive_wait(ctr, 2 * nwg);
  grid_leave(ctr, 3 * nwg);
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  if (rl >= RL) return;
  float mean[W], istd[W], sc[W], bt[W], gi[W], m1[W], m2[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int ch = cg * W + k;
    mean[k] = stats[ch]; istd[k] = stats[c + ch];
    sc[k] = istd[k] * (relu == 2 ? gamma[ch] : 0.f);
    bt[k] = relu == 2 ? beta[ch] : 0.f;
    gi[k] = gamma[ch] * istd[k];
    m1[k] = ld_coh(sums + ch) * inv_n; m2[k] = ld_coh(sums + c + ch) * inv_n;
  }
  constexpr int U = W == 8 ? 1 : 2;
  for (int64_t r = r1 - 1 - rl; r >= r0; r -= (int64_t)U * RL) {
    float xv[U][W], gv[U][W], yv[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ru = r - (int64_t)u * RL;
      if (ru >= r0) {
        Vec<T>::load(x + ru * c + cg * W, xv[u]);
        Vec<T>::load(dy + ru * dy_ld + cg * W, gv[u]);
        if (relu == 1) Vec<T>::load(y + ru * y_ld + cg * W, yv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t ru = r - (int64_t)u * RL;
      if (ru >= r0) {
        if (relu == 1) {
#pragma unroll
          for (int k = 0; k < W; ++k) gv[u][k] = yv[u][k] > 0.f ? gv[u][k] : 0.f;
        } else if (relu == 2) {
#pragma unroll
          for (int k = 0; k < W; ++k) gv[u][k] = ((xv[u][k] - mean[k]) * sc[k] + bt[k]) > 0.f ? gv[u][k] : 0.f;
        }
        if (dres) Vec<T>::store(dres + ru * c + cg * W, gv[u]);
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const float xh = (xv[u][k] - mean[k]) * istd[k];
          xv[u][k] = gi[k] * (gv[u][k] - m1[k] - xh * m2[k]);
        }
        Vec<T>::store(dx + ru * c + cg * W, xv[u]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ two launches, no barrier
// Small layers (the coarse levels: a few thousand to ~100 k rows) are latency chains, not bandwidth: three dependent launches
// per direction, or ONE launch with two hand-rolled grid barriers whose workgroups must all be resident before the first may
// pass (round 4: 53 us per backward launch in the step, 8 % of the HBM peak; the weight-gradient kernel on the side stream owns
// CUs for a millisecond at a time).  Here a direction is TWO ordinary launches: the column reduction into <= kFoldParts partial
// rows, and an apply kernel whose every workgroup first folds those partial rows itself -- the threads of a workgroup that share
// a channel group split the rows between them (fixed assignment, double accumulation, combined through LDS in lane order: every
// workgroup computes bit-identical statistics) -- and then streams its rows.  No inter-workgroup synchronisation, no fold
// launch; the redundant fold reads (workgroups x parts x 2C floats) stay in L2.  Workgroup 0 also writes what the fold kernels
// write: saved statistics, running statistics, num_batches_tracked / d gamma, d beta.
constexpr int kFoldParts = 64;      // partial rows (upper bound; knob BN_FOLD_PARTS)

// -> this thread's channel-group totals of the two column sums, as doubles
template <int W>
__device__ inline void fold_in_block(const float *__restrict__ scratch, int nparts, int c, int G, int RL, int cg, int rl,
                                     double (&t0)[W], double (&t1)[W]) {
  __shared__ double fred[2][kNT][W];
  double a0[W], a1[W];
#pragma unroll
  for (int i = 0; i < W; ++i) a0[i] = a1[i] = 0.0;
  if (rl < RL) {
    for (int b = rl; b < nparts; b += RL) {
      const float *row = scratch + (int64_t)b * 2 * c + cg * W;
#pragma unroll
      for (int i = 0; i < W; ++i) { a0[i] += (double)row[i]; a1[i] += (double)row[c + i]; }
    }
  }
#pragma unroll
  for (int i = 0; i < W; ++i) { fred[0][threadIdx.x][i] = a0[i]; fred[1][threadIdx.x][i] = a1[i]; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < W; ++i) { t0[i] = 0.0; t1[i] = 0.0; }
  if (rl < RL) {
    const int lim = RL < nparts ? RL : nparts;
    for (int j = 0; j < lim; ++j) {
#pragma unroll
      for (int i = 0; i < W; ++i) { t0[i] += fred[0][j * G + cg][i]; t1[i] += fred[1][j * G + cg][i]; }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kNT) void k_bn_apply_fold(const T *__restrict__ x, const T *__restrict__ res, int64_t n, int c,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                       float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                                                       long long *__restrict__ nbt, float *__restrict__ stats,
                                                       const float *__restrict__ scratch, int nparts, int relu, T *__restrict__ y,
                                                       int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  double t0[W], t1[W];
  fold_in_block<W>(scratch, nparts, c, G, RL, cg, rl, t0, t1);
  if (rl >= RL) return;
  float piv[W], sc[W], mean[W], bt[W];
  if (n > 0) Vec<T>::load(x + cg * W, piv);          // the pivot of k_colreduce<T, 0>: row 0
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int ch = cg * W + k;
    const double dm = n > 0 ? t0[k] / (double)n : 0.0;
    double var = n > 0 ? t1[k] / (double)n - dm * dm : 0.0;
    if (var < 0.0) var = 0.0;
    const double m = (n > 0 ? (double)piv[k] : 0.0) + dm;
    const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
    mean[k] = mf; sc[k] = isf * gamma[ch]; bt[k] = beta[ch];
    if (blockIdx.x == 0 && rl == 0) {
      stats[ch] = mf; stats[c + ch] = isf;
      if (running_mean) {
        const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * m);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
      }
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  const int64_t stride = (int64_t)gridDim.x * RL;
  for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < n; r += 2 * stride) {
    const int64_t r2 = r + stride;
    const bool two = r2 < n;
    float xa[W], xb[W], ra[W], rb[W];
    Vec<T>::load(x + r * c + cg * W, xa);
    if (two) Vec<T>::load(x + r2 * c + cg * W, xb);
    if (res) { Vec<T>::load(res + r * c + cg * W, ra); if (two) Vec<T>::load(res + r2 * c + cg * W, rb); }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      float o = (xa[k] - mean[k]) * sc[k] + bt[k];   // the exact expression of k_bn_apply (the backward recomputes the mask from it)
      if (res) o += ra[k];
      xa[k] = (relu && o < 0.f) ? 0.f : o;
    }
    Vec<T>::store(y + r * y_ld + cg * W, xa);
    if (two) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        float o = (xb[k] - mean[k]) * sc[k] + bt[k];
        if (res) o += rb[k];
        xb[k] = (relu && o < 0.f) ? 0.f : o;
      }
      Vec<T>::store(y + r2 * y_ld + cg * W, xb);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kNT) void k_bn_bwd_apply_fold(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy,
                                                           int64_t n, int c, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ stats,
                                                           const float *__restrict__ scratch, int nparts, float inv_n, int relu,
                                                           T *__restrict__ dx, T *__restrict__ dres, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta, int64_t dy_ld, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  const int G = c / W, RL = kNT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  double t0[W], t1[W];
  fold_in_block<W>(scratch, nparts, c, G, RL, cg, rl, t0, t1);
  if (rl >= RL) return;
  float mean[W], istd[W], sc[W], bt[W], gi[W], m1[W], m2[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int ch = cg * W + k;
    const float s = (float)t0[k], ss = (float)t1[k];      // what k_fold_bwd stores (and k_bn_bwd_apply then reads)
    if (blockIdx.x == 0 && rl == 0) { dbeta[ch] = s; dgamma[ch] = ss; }
    mean[k] = stats[ch]; istd[k] = stats[c + ch];
    sc[k] = istd[k] * (relu == 2 ? gamma[ch] : 0.f);
    bt[k] = relu == 2 ? beta[ch] : 0.f;
    gi[k] = gamma[ch] * istd[k];
    m1[k] = s * inv_n; m2[k] = ss * inv_n;
  }
  const int64_t stride = (int64_t)gridDim.x * RL;
  for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < n; r += stride) {
    const int64_t o = r * c + cg * W;
    float xv[W], gv[W];
    Vec<T>::load(x + o, xv);
    Vec<T>::load(dy + r * dy_ld + cg * W, gv);
    if (relu == 1) {
      float yv[W];
      Vec<T>::load(y + r * y_ld + cg * W, yv);
#pragma unroll
      for (int k = 0; k < W; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    } else if (relu == 2) {
#pragma unroll
      for (int k = 0; k < W; ++k) gv[k] = ((xv[k] - mean[k]) * sc[k] + bt[k]) > 0.f ? gv[k] : 0.f;  // same expression as k_bn_apply
    }
    if (dres) Vec<T>::store(dres + o, gv);
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float xh = (xv[k] - mean[k]) * istd[k];
      xv[k] = gi[k] * (gv[k] - m1[k] - xh * m2[k]);
    }
    Vec<T>::store(dx + o, xv);
  }
}

// partial rows / apply workgroups of the two-launch path (knobs BN_FOLD_PARTS / BN_FOLD_GRID)
inline bool bn_fold_on(int64_t tensor_bytes) { return tune(T_BN_FOLD) != 0 && tensor_bytes <= (tune(T_BN_FOLD_MAX_MB) << 20); }
inline int fold_parts(int64_t n, int64_t *rows_per_block) {
  int64_t cap = tune(T_BN_FOLD_PARTS);
  if (cap < 1 || cap > kFoldParts) cap = kFoldParts;
  int64_t nb = (n + 127) / 128;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  int64_t rpb = (n + nb - 1) / nb;
  if (rpb < 1) rpb = 1;
  *rows_per_block = rpb;
  nb = (n + rpb - 1) / rpb;
  return (int)(nb > 0 ? nb : 1);
}
inline int fold_grid(int64_t n, int c, int W) {
  int64_t cap = tune(T_BN_FOLD_GRID);
  if (cap < 1) cap = 256;
  const int64_t total = n * (int64_t)(c / W);
  int64_t g = (total + 2 * kNT - 1) / (2 * kNT);       // two rows per thread and sweep in the forward kernel
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

// barrier counters of the fused kernels: a ring of slots per device, zeroed once; a kernel leaves its slot at zero
constexpr int kCtrSlots = 256;
// `s`: the stream of the launch that asks.  The ring is zeroed ONCE, asynchronously on the first asker's stream; until that memset
// is known to have completed every asker's stream is ordered behind it through an event -- no host synchronisation anywhere
// (round 4 blocked the host in hipDeviceSynchronize at the first BatchNorm of a process).
inline unsigned *fused_counter(hipStream_t s) {
  static unsigned *ring[64] = {nullptr};
  static hipEvent_t zeroed[64] = {nullptr};
  static std::atomic<bool> settled[64];
  static std::atomic<unsigned> next[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!settled[dev].load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(mu);
    if (!ring[dev]) {
      unsigned *p = nullptr;
      if (hipMalloc(&p, kCtrSlots * sizeof(unsigned)) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&zeroed[dev], hipEventDisableTiming) != hipSuccess ||
          hipMemsetAsync(p, 0, kCtrSlots * sizeof(unsigned), s) != hipSuccess || hipEventRecord(zeroed[dev], s) != hipSuccess)
        return nullptr;
      ring[dev] = p;
    } else if (hipEventQuery(zeroed[dev]) == hipSuccess) {
      settled[dev].store(true, std::memory_order_release);
    } else {
      (void)hipGetLastError();                                   // hipErrorNotReady is not an error
      if (hipStreamWaitEvent(s, zeroed[dev], 0) != hipSuccess) return nullptr;
    }
  }
  return ring[dev] + (next[dev].fetch_add(1) % kCtrSlots);
}
inline bool bn_fused_on(int64_t tensor_bytes, bool forward = false) {
  const bool on = tune(T_BN_FUSED) != 0;   // A/B knob: 0 = three launches
  // above ~24 MB a direction is bandwidth-bound and the three-launch path's 4096-workgroup apply streams faster than 512
  // resident workgroups can (1.2 M rows x 96 ch bf16 forward: 0.135 ms vs 0.181 ms fused); below, launches dominate
  int64_t max_mb = tune(T_BN_FUSED_MAX_MB);
  if (forward && tune(T_BN_FUSED_FWD_MAX_MB) < max_mb) max_mb = tune(T_BN_FUSED_FWD_MAX_MB);
  return on && tensor_bytes <= (max_mb << 20);
}
// Workgroups a grid-barrier kernel may be launched with: every one of them must be RESIDENT at the same time, or the resident
// ones spin on the barrier for workgroups the dispatcher can never place (a hard GPU hang).  The bound is the kernel's own
// occupancy on THIS device x its CU count (a CPX partition of the chip has 32-38 CUs, not 256), halved for headroom against
// another stream's kernels holding CUs, cached per (device, kernel).  Below 16 the fused path is not worth it: 0 = use the
// three-launch path.  What this cannot see: HSA_CU_MASK-style external masks and other PROCESSES on the same GPU -- those
// setups must run with the tuning knob BN_FUSED=0 (bench.py --same-device does).
inline int fused_cap(const void *kernel) {
  static std::mutex mu;
  static std::vector<std::pair<std::pair<int, const void *>, int>> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(mu);
  for (auto &e : cache)
    if (e.first.first == dev && e.first.second == kernel) return e.second;
  int per_cu = 0, cus = 0, cap = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kNT, 0) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
    cap = per_cu * cus / 2;
  else
    (void)hipGetLastError();
  if (cap > kFusedMaxBlocks) cap = kFusedMaxBlocks;
  if (cap < 16) cap = 0;
  cache.push_back({{dev, kernel}, cap});
  return cap;
}
inline int fused_blocks(int64_t n, int64_t *rows_per_block, int cap) {
  const int cap_env = (int)tune(T_BN_FUSED_BLOCKS);   // tuning knob: 0 = the device bound
  if (cap_env > 0 && cap_env < cap) cap = cap_env;
  int64_t nb = (n + 127) / 128;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  int64_t rpb = (n + nb - 1) / nb;
  if (rpb < 1) rpb = 1;
  *rows_per_block = rpb;
  nb = (n + rpb - 1) / rpb;
  return (int)(nb > 0 ? nb : 1);
}

// statistics source: either the column reduction over x, or the conv epilogue's per-tile partial rows
template <typename T>
int stats_partials(const T *x, int64_t n, int c, const float *partials, int partial_rows, float *scratch, hipStream_t s, int *nb_out) {
  if (partials && partial_rows > 0) {
    int nb = partial_rows < 128 ? partial_rows : 128;
    const int rpb = (partial_rows + nb - 1) / nb;
    nb = (partial_rows + rpb - 1) / rpb;
    LGS_KLAUNCH(k_partial_reduce, nb, 256, 0, s, partials, partial_rows, 2 * c, rpb, scratch);
    *nb_out = nb;
    return 0;
  }
  int64_t rpb;
  const int nb = reduce_blocks(n, &rpb);
  LGS_KLAUNCH((k_colreduce<T, 0>), nb, kNT, 0, s, x, (const T *)nullptr, (const T *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, n, c, 0,
                     rpb, scratch, (int64_t)c, (int64_t)c);
  *nb_out = nb;
  return 0;
}

template <typename T>
int bn_forward_t(const void *xv, int64_t n, int c, const float *gamma, const float *beta, float eps, float momentum,
                 float *rm, float *rv, long long *nbt, const void *res, int relu, void *yv, float *stats, void *workspace,
                 hipStream_t s, const float *partials, int partial_rows, const float *pivot, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  LGS_REQUIRE(c % W == 0 && c / W <= kNT && c <= 2048, "lgs_bn_forward: channel count unsupported");
  int nb = 0;
  float *scratch = reinterpret_cast<float *>(workspace);  // caller-owned: no allocator call (and no implicit sync) here
  const T *x = reinterpret_cast<const T *>(xv);
  const int pm = (partials && partial_rows > 0) ? 1 : 0;
  if (!pm && n > 0 && bn_fold_on(n * (int64_t)c * (int64_t)sizeof(T))) {
    int64_t rpb;
    const int parts = fold_parts(n, &rpb);
    LGS_KLAUNCH((k_colreduce<T, 0>), parts, kNT, 0, s, x, (const T *)nullptr, (const T *)nullptr, (const float *)nullptr, (const float *)nullptr,
                (const float *)nullptr, n, c, 0, rpb, scratch, (int64_t)c, (int64_t)c);
    LGS_KLAUNCH((k_bn_apply_fold<T>), fold_grid(n, c, W), kNT, 0, s, x, reinterpret_cast<const T *>(res), n, c, gamma, beta, eps, momentum,
                rm, rv, nbt, stats, scratch, parts, relu, reinterpret_cast<T *>(yv), y_ld);
    LGS_HIP(hipGetLastError());
    return 0;
  }
  const int fcap = bn_fused_on(n * (int64_t)c * (int64_t)sizeof(T), true) ? fused_cap(reinterpret_cast<const void *>(&k_bn_fwd_fused<T>)) : 0;
  if (fcap > 0) {
    unsigned *ctr = fused_counter(s);
    LGS_REQUIRE(ctr != nullptr, "lgs_bn_forward: could not allocate the grid-barrier counters");
    int64_t rpb;
    const int grid = fused_blocks(n, &rpb, fcap);
    int prpb = 0, nfold = grid;
    if (pm) {
      int pnb = partial_rows < 128 ? partial_rows : 128;
      prpb = (partial_rows + pnb - 1) / pnb;
      nfold = (partial_rows + prpb - 1) / prpb;
    }
    LGS_KLAUNCH((k_bn_fwd_fused<T>), grid, kNT, 0, s, x, reinterpret_cast<const T *>(res), n, c, gamma, beta, eps, momentum, rm, rv, nbt,
                       stats, relu, reinterpret_cast<T *>(yv), y_ld, scratch, pm ? partials : (const float *)nullptr, partial_rows, prpb, nfold,
                       pivot, rpb, ctr);
    LGS_HIP(hipGetLastError());
    return 0;
  }
  stats_partials<T>(x, n, c, partials, partial_rows, scratch, s, &nb);
  LGS_KLAUNCH((k_fold_fwd<T>), (c + kFoldCh - 1) / kFoldCh, 256, 0, s, scratch, x, nb, c, n, eps, momentum, rm, rv, nbt, stats, pm, pivot);
  int64_t total = n * (int64_t)(c / W);
  if (total > 0) {
    int grid = (int)((total + kNT - 1) / kNT < 4096 ? (total + kNT - 1) / kNT : 4096);
    LGS_KLAUNCH((k_bn_apply<T>), grid, kNT, 0, s, x, reinterpret_cast<const T *>(res), n, c, gamma, beta, stats, relu,
                       reinterpret_cast<T *>(yv), y_ld);
  }
  LGS_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int bn_backward_t(const void *xv, const void *yv, const void *dyv, int64_t n, int c, const float *gamma, const float *beta,
                  const float *stats, int relu, void *dxv, void *dresv, float *dgamma, float *dbeta, void *workspace,
                  hipStream_t s, int64_t dy_ld, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  LGS_REQUIRE(c % W == 0 && c / W <= kNT && c <= 2048, "lgs_bn_backward: channel count unsupported");
  int64_t rpb;
  int nb = reduce_blocks(n, &rpb);
  float *scratch = reinterpret_cast<float *>(workspace);
  float *sums = scratch + (size_t)2 * c * nb;
  const T *x = reinterpret_cast<const T *>(xv), *y = reinterpret_cast<const T *>(yv), *dy = reinterpret_cast<const T *>(dyv);
  if (n > 0 && bn_fold_on(n * (int64_t)c * (int64_t)sizeof(T))) {
    int64_t prpb;
    const int parts = fold_parts(n, &prpb);
    LGS_KLAUNCH((k_colreduce<T, 1>), parts, kNT, 0, s, x, y, dy, stats, gamma, beta, n, c, relu, prpb, scratch, dy_ld, y_ld);
    LGS_KLAUNCH((k_bn_bwd_apply_fold<T>), fold_grid(n, c, W), kNT, 0, s, x, y, dy, n, c, gamma, beta, stats, scratch, parts, 1.f / (float)n, relu,
                reinterpret_cast<T *>(dxv), reinterpret_cast<T *>(dresv), dgamma, dbeta, dy_ld, y_ld);
    LGS_HIP(hipGetLastError());
    return 0;
  }
  const int fcap = bn_fused_on(n * (int64_t)c * (int64_t)sizeof(T)) ? fused_cap(reinterpret_cast<const void *>(&k_bn_bwd_fused<T>)) : 0;
  if (fcap > 0) {
    unsigned *ctr = fused_counter(s);
    LGS_REQUIRE(ctr != nullptr, "lgs_bn_backward: could not allocate the grid-barrier counters");
    int64_t frpb;
    const int grid = fused_blocks(n, &frpb, fcap);
    LGS_KLAUNCH((k_bn_bwd_fused<T>), grid, kNT, 0, s, x, y, dy, n, c, gamma, beta, stats, relu, reinterpret_cast<T *>(dxv),
                       reinterpret_cast<T *>(dresv), dgamma, dbeta, scratch, scratch + (size_t)2 * c * grid, dy_ld, y_ld, frpb,
                       n > 0 ? 1.f / (float)n : 0.f, ctr);
    LGS_HIP(hipGetLastError());
    return 0;
  }
  LGS_KLAUNCH((k_colreduce<T, 1>), nb, kNT, 0, s, x, y, dy, stats, gamma, beta, n, c, relu, rpb, scratch, dy_ld, y_ld);
  LGS_KLAUNCH(k_fold_bwd, (c + kFoldCh - 1) / kFoldCh, 256, 0, s, scratch, nb, c, dgamma, dbeta, sums);
  int64_t total = n * (int64_t)(c / W);
  if (total > 0) {
    int grid = (int)((total + kNT - 1) / kNT < 4096 ? (total + kNT - 1) / kNT : 4096);
    LGS_KLAUNCH((k_bn_bwd_apply<T>), grid, kNT, 0, s, x, y, dy, n, c, gamma, beta, stats, sums, n > 0 ? 1.f / (float)n : 0.f, relu, reinterpret_cast<T *>(dxv),
                       reinterpret_cast<T *>(dresv), dy_ld, (const float *)nullptr, y_ld);
  }
  LGS_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int bn_stats_t(const void *xv, int64_t n, int c, float *mean_m2, void *workspace, hipStream_t s, const float *partials,
               int partial_rows, const float *pivot) {
  constexpr int W = Vec<T>::W;
  LGS_REQUIRE(c % W == 0 && c / W <= kNT && c <= 2048, "lgs_bn_stats: channel count unsupported");
  int nb = 0;
  float *scratch = reinterpret_cast<float *>(workspace);
  const T *x = reinterpret_cast<const T *>(xv);
  const int pm = (partials && partial_rows > 0) ? 1 : 0;
  stats_partials<T>(x, n, c, partials, partial_rows, scratch, s, &nb);
  LGS_KLAUNCH((k_fold_stats<T>), (c + kFoldCh - 1) / kFoldCh, 256, 0, s, scratch, x, nb, c, n, mean_m2, pm, pivot);
  LGS_HIP(hipGetLastError());
  return 0;
}
template <typename T>
int bn_apply_t(const void *xv, int64_t n, int c, const float *gamma, const float *beta, const float *stats, const void *res,
               int relu, void *yv, hipStream_t s, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  LGS_REQUIRE(c % W == 0 && c / W <= kNT, "lgs_bn_apply: channel count unsupported");
  int64_t total = n * (int64_t)(c / W);
  if (total > 0) {
    int grid = (int)((total + kNT - 1) / kNT < 4096 ? (total + kNT - 1) / kNT : 4096);
    LGS_KLAUNCH((k_bn_apply<T>), grid, kNT, 0, s, reinterpret_cast<const T *>(xv), reinterpret_cast<const T *>(res), n, c,
                       gamma, beta, stats, relu, reinterpret_cast<T *>(yv), y_ld);
  }
  LGS_HIP(hipGetLastError());
  return 0;
}
template <typename T>
int bn_bwd_reduce_t(const void *xv, const void *yv, const void *dyv, int64_t n, int c, const float *gamma, const float *beta,
                    const float *stats, int relu, float *sums, float *dgamma, float *dbeta, void *workspace, hipStream_t s,
                    int64_t dy_ld, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  LGS_REQUIRE(c % W == 0 && c / W <= kNT && c <= 2048, "lgs_bn_backward_reduce: channel count unsupported");
  int64_t rpb;
  int nb = reduce_blocks(n, &rpb);
  float *scratch = reinterpret_cast<float *>(workspace);
  float *tmp = scratch + (size_t)2 * c * nb;  // dgamma/dbeta land here when the caller does not want them
  LGS_KLAUNCH((k_colreduce<T, 1>), nb, kNT, 0, s, reinterpret_cast<const T *>(xv), reinterpret_cast<const T *>(yv),
                     reinterpret_cast<const T *>(dyv), stats, gamma, beta, n, c, relu, rpb, scratch, dy_ld, y_ld);
  LGS_KLAUNCH(k_fold_bwd, (c + kFoldCh - 1) / kFoldCh, 256, 0, s, scratch, nb, c, dgamma ? dgamma : tmp + c, dbeta ? dbeta : tmp, sums);
  LGS_HIP(hipGetLastError());
  return 0;
}
template <typename T>
int bn_bwd_apply_t(const void *xv, const void *yv, const void *dyv, int64_t n, int c, const float *gamma, const float *beta,
                   const float *stats, const float *sums, float inv_n_total, const float *inv_n_dev, int relu, void *dxv, void *dresv,
                   hipStream_t s, int64_t dy_ld, int64_t y_ld) {
  constexpr int W = Vec<T>::W;
  LGS_REQUIRE(c % W == 0 && c / W <= kNT, "lgs_bn_backward_apply: channel count unsupported");
  int64_t total = n * (int64_t)(c / W);
  if (total > 0) {
    int grid = (int)((total + kNT - 1) / kNT < 4096 ? (total + kNT - 1) / kNT : 4096);
    LGS_KLAUNCH((k_bn_bwd_apply<T>), grid, kNT, 0, s, reinterpret_cast<const T *>(xv), reinterpret_cast<const T *>(yv),
                       reinterpret_cast<const T *>(dyv), n, c, gamma, beta, stats, sums, inv_n_total, relu, reinterpret_cast<T *>(dxv),
                       reinterpret_cast<T *>(dresv), dy_ld, inv_n_dev, y_ld);
  }
  LGS_HIP(hipGetLastError());
  return 0;
}

// row stride of a [n, c] operand that may be a column slice of a wider row-major buffer: 0 = c; rows must start 16-byte aligned
inline bool stride_ok(const void *p, int64_t ld, int c, int dtype) {
  return ld >= c && ld % (dtype == LGS_BF16 ? 8 : 4) == 0 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

}  // namespace lgs

using namespace lgs;

extern "C" {

int lgs_bn_stats(const void *x, int64_t n, int c, float *mean_m2, int dtype, void *workspace, const float *conv_partials,
                 int conv_partial_rows, const float *pivot, void *stream) {
  LGS_REQUIRE(x && mean_m2 && workspace, "lgs_bn_stats: null argument");
  if (dtype == LGS_F32) return bn_stats_t<float>(x, n, c, mean_m2, workspace, (hipStream_t)stream, conv_partials, conv_partial_rows, pivot);
  if (dtype == LGS_BF16) return bn_stats_t<bf16_t>(x, n, c, mean_m2, workspace, (hipStream_t)stream, conv_partials, conv_partial_rows, pivot);
  LGS_REQUIRE(false, "lgs_bn_stats: unknown dtype");
}
int lgs_bn_apply(const void *x, int64_t n, int c, const float *gamma, const float *beta, const float *stats,
                 const void *residual, int relu, void *y, int dtype, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(x && y && gamma && beta && stats, "lgs_bn_apply: null argument");
  const int64_t y_ld = y_row_stride > 0 ? y_row_stride : c;
  LGS_REQUIRE(stride_ok(y, y_ld, c, dtype), "lgs_bn_apply: y rows must start 16-byte aligned (row stride a multiple of 16 bytes)");
  if (dtype == LGS_F32) return bn_apply_t<float>(x, n, c, gamma, beta, stats, residual, relu, y, (hipStream_t)stream, y_ld);
  if (dtype == LGS_BF16) return bn_apply_t<bf16_t>(x, n, c, gamma, beta, stats, residual, relu, y, (hipStream_t)stream, y_ld);
  LGS_REQUIRE(false, "lgs_bn_apply: unknown dtype");
}
int lgs_bn_sync_combine(const float *all_stats, int world, int c, float eps, float momentum, float *running_mean,
                        float *running_var, int64_t *num_batches_tracked, float *stats, float *inv_n_total, void *stream) {
  LGS_REQUIRE(all_stats && stats && world > 0 && c > 0, "lgs_bn_sync_combine: bad argument");
  LGS_KLAUNCH(k_sync_combine, (unsigned)((c + 127) / 128), 128, 0, (hipStream_t)stream, all_stats, world, c, eps, momentum,
                     running_mean, running_var, reinterpret_cast<long long *>(num_batches_tracked), stats, inv_n_total);
  LGS_HIP(hipGetLastError());
  return 0;
}
int lgs_bn_backward_reduce(const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                           const float *beta, const float *stats, int relu, float *sums, float *dgamma, float *dbeta, int dtype,
                           void *workspace, int64_t dy_row_stride, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(x && dy && stats && sums && workspace, "lgs_bn_backward_reduce: null argument");
  const int64_t dy_ld = dy_row_stride > 0 ? dy_row_stride : c, y_ld = y_row_stride > 0 ? y_row_stride : c;
  LGS_REQUIRE(stride_ok(dy, dy_ld, c, dtype) && (!y || stride_ok(y, y_ld, c, dtype)),
              "lgs_bn_backward_reduce: dy / y rows must start 16-byte aligned (row stride a multiple of 16 bytes)");
  LGS_REQUIRE(relu != 1 || y, "lgs_bn_backward_reduce: relu mode 1 needs the forward output");
  LGS_REQUIRE(relu != 2 || (gamma && beta), "lgs_bn_backward_reduce: relu mode 2 needs gamma and beta");
  if (dtype == LGS_F32) return bn_bwd_reduce_t<float>(x, y, dy, n, c, gamma, beta, stats, relu, sums, dgamma, dbeta, workspace, (hipStream_t)stream, dy_ld, y_ld);
  if (dtype == LGS_BF16) return bn_bwd_reduce_t<bf16_t>(x, y, dy, n, c, gamma, beta, stats, relu, sums, dgamma, dbeta, workspace, (hipStream_t)stream, dy_ld, y_ld);
  LGS_REQUIRE(false, "lgs_bn_backward_reduce: unknown dtype");
}
int lgs_bn_backward_apply(const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                          const float *beta, const float *stats, const float *sums, float inv_n_total,
                          const float *inv_n_device, int relu, void *dx, void *dresidual, int dtype, int64_t dy_row_stride,
                          int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(x && dy && dx && gamma && stats && sums, "lgs_bn_backward_apply: null argument");
  const int64_t dy_ld = dy_row_stride > 0 ? dy_row_stride : c, y_ld = y_row_stride > 0 ? y_row_stride : c;
  LGS_REQUIRE(stride_ok(dy, dy_ld, c, dtype) && (!y || stride_ok(y, y_ld, c, dtype)),
              "lgs_bn_backward_apply: dy / y rows must start 16-byte aligned (row stride a multiple of 16 bytes)");
  LGS_REQUIRE(relu != 1 || y, "lgs_bn_backward_apply: relu mode 1 needs the forward output");
  LGS_REQUIRE(relu != 2 || beta, "lgs_bn_backward_apply: relu mode 2 needs beta");
  if (dtype == LGS_F32) return bn_bwd_apply_t<float>(x, y, dy, n, c, gamma, beta, stats, sums, inv_n_total, inv_n_device, relu, dx, dresidual, (hipStream_t)stream, dy_ld, y_ld);
  if (dtype == LGS_BF16) return bn_bwd_apply_t<bf16_t>(x, y, dy, n, c, gamma, beta, stats, sums, inv_n_total, inv_n_device, relu, dx, dresidual, (hipStream_t)stream, dy_ld, y_ld);
  LGS_REQUIRE(false, "lgs_bn_backward_apply: unknown dtype");
}

int64_t lgs_bn_workspace_bytes(int64_t n, int c) {
  int64_t rpb;
  int nb = reduce_blocks(n, &rpb);
  if (nb < 128) nb = 128;   // k_partial_reduce folds conv-epilogue partial rows into <= 128 rows
  return (int64_t)sizeof(float) * 2 * c * (nb + 2) + 256;
}

int lgs_bn_forward(const void *x, int64_t n, int c, const float *gamma, const float *beta, float eps, float momentum,
                   float *running_mean, float *running_var, int64_t *num_batches_tracked, const void *residual, int relu,
                   void *y, float *stats, int dtype, void *workspace, const float *conv_partials, int conv_partial_rows,
                   const float *pivot, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(x && y && gamma && beta && stats && workspace, "lgs_bn_forward: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t y_ld = y_row_stride > 0 ? y_row_stride : c;
  LGS_REQUIRE(y_ld >= c && y_ld % (dtype == LGS_BF16 ? 8 : 4) == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0,
              "lgs_bn_forward: y rows must start 16-byte aligned (row stride a multiple of 16 bytes)");
  if (dtype == LGS_F32) return bn_forward_t<float>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var, reinterpret_cast<long long *>(num_batches_tracked), residual, relu, y, stats, workspace, s, conv_partials, conv_partial_rows, pivot, y_ld);
  if (dtype == LGS_BF16) return bn_forward_t<bf16_t>(x, n, c, gamma, beta, eps, momentum, running_mean, running_var, reinterpret_cast<long long *>(num_batches_tracked), residual, relu, y, stats, workspace, s, conv_partials, conv_partial_rows, pivot, y_ld);
  LGS_REQUIRE(false, "lgs_bn_forward: unknown dtype");
}

int lgs_bn_backward(const void *x, const void *y, const void *dy, int64_t dy_row_stride, int64_t n, int c, const float *gamma,
                    const float *beta, const float *stats, int relu, void *dx, void *dresidual, float *dgamma, float *dbeta,
                    int dtype, void *workspace, int64_t y_row_stride, void *stream) {
  LGS_REQUIRE(x && dy && dx && gamma && stats && dgamma && dbeta && workspace, "lgs_bn_backward: null argument");
  const int64_t dy_ld = dy_row_stride > 0 ? dy_row_stride : c;
  LGS_REQUIRE(dy_ld >= c && dy_ld % (dtype == LGS_BF16 ? 8 : 4) == 0 &&
                  (reinterpret_cast<uintptr_t>(dy) & 15u) == 0,
              "lgs_bn_backward: dy rows must start 16-byte aligned (row stride a multiple of 16 bytes)");
  LGS_REQUIRE(relu != 1 || y, "lgs_bn_backward: relu mode 1 needs the forward output");
  LGS_REQUIRE(relu != 2 || beta, "lgs_bn_backward: relu mode 2 needs beta");
  hipStream_t s = (hipStream_t)stream;
  const int64_t y_ld = y_row_stride > 0 ? y_row_stride : c;
  LGS_REQUIRE(!y || (y_ld >= c && y_ld % (dtype == LGS_BF16 ? 8 : 4) == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0),
              "lgs_bn_backward: y rows must start 16-byte aligned (row stride a multiple of 16 bytes)");
  if (dtype == LGS_F32) return bn_backward_t<float>(x, y, dy, n, c, gamma, beta, stats, relu, dx, dresidual, dgamma, dbeta, workspace, s, dy_ld, y_ld);
  if (dtype == LGS_BF16) return bn_backward_t<bf16_t>(x, y, dy, n, c, gamma, beta, stats, relu, dx, dresidual, dgamma, dbeta, workspace, s, dy_ld, y_ld);
  LGS_REQUIRE(false, "lgs_bn_backward: unknown dtype");
}

}  // extern "C"
