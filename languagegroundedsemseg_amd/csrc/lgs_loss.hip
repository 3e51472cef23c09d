// lgs_loss.hip -- fused softmax cross-entropy (forward + gradient in one pass) over [N, C] logits, gfx950.
//
// Replaces nn.CrossEntropyLoss(ignore_index=-1) on the [N,200] logits of the fine-tune step
//   /root/reference/lib/train_test/pl_BaselineTrainer.py:94-99,350
// HBM-bound: logits are read once (16-byte loads, half a wavefront per row), the per-row loss and the
// gradient (softmax - onehot) * scale are written in the same pass; ignored rows produce 0 / zeros.
#include "lgs_common.h"

namespace lgs {

template <typename T> struct LVec;
template <> struct LVec<float> {
  static constexpr int W = 4;
  __device__ static void load(const float *p, float (&v)[4]) { float4 x = *reinterpret_cast<const float4 *>(p); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
  __device__ static void store(float *p, const float (&v)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct LVec<bf16_t> {
  static constexpr int W = 8;
  __device__ static void load(const bf16_t *p, float (&v)[8]) {
    uint4 x = *reinterpret_cast<const uint4 *>(p);
    uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_to_f32((uint16_t)(w[i] & 0xffff)); v[2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16)); }
  }
  __device__ static void store(bf16_t *p, const float (&v)[8]) {
    uint4 x;
    x.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    x.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    x.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    x.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = x;
  }
};

constexpr int kMaxChunks = 4;  // 16-byte chunks per lane per row: C <= 32 * 4 * W
__device__ inline void st_elem(float *p, float v) { *p = v; }
__device__ inline void st_elem(bf16_t *p, float v) { *p = f32_to_bf16(v); }

// Half a wavefront per row; a half-wave takes R consecutive rows and issues the loads of ALL of them before it touches the first
// (round 6: with one 400-byte row in flight per half-wave the 1.2 M x 200 launch ran at 2.8 TB/s -- 32 waves per CU x 800 bytes is
// not enough outstanding traffic; Q = 16-byte chunks per lane and row, Q x R = 4 keeps the register count where it was).
// The arithmetic per row is unchanged (same reduction tree, same order): results are bit-identical to the one-row kernel.
template <typename T, int Q, int R>
__global__ __launch_bounds__(256) void k_ce_fwd_bwd(const T *__restrict__ logits, int64_t n, int c, const int64_t *__restrict__ labels,
                                                    int64_t ignore_index, const float *__restrict__ scale_ptr,
                                                    const float *__restrict__ row_scale, float *__restrict__ loss_rows,
                                                    T *__restrict__ dlogits) {
  constexpr int W = LVec<T>::W;
  const int lane = threadIdx.x & 31;  // half-wave per row group
  const int64_t row0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5) * R;
  if (row0 >= n) return;
  // class counts that are not a multiple of the 16-byte width (e.g. the 20 ScanNet classes in bf16) take
  // element-wise loads / stores for their rows: kernel-uniform branch, padding lanes hold -inf -> exp = 0
  const bool vec = (c % W) == 0;
  const int nchunk = (c + W - 1) / W;
  float v[R][Q][W];
  int64_t labs[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + r;
    labs[r] = row < n ? labels[row] : -1;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int ch = q * 32 + lane;
      if (ch < nchunk && row < n) {
        if (vec) {
          LVec<T>::load(logits + row * c + ch * W, v[r][q]);
        } else {
#pragma unroll
          for (int i = 0; i < W; ++i) v[r][q][i] = ch * W + i < c ? ld_elem(logits + row * c + ch * W + i) : -3.0e38f;
        }
      }
    }
  }
  const float scale0 = dlogits ? *scale_ptr : 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + r;
    if (row >= n) break;              // (uniform per half-wave: the shuffles below stay inside it)
    const int64_t lab = labs[r];
    const bool ignored = (lab == ignore_index) || lab < 0 || lab >= c;
    const int lab32 = ignored ? -1 : (int)lab;          // (32-bit compares against compile-time element offsets below)
    float mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int ch = q * 32 + lane;
      if (ch < nchunk) {
#pragma unroll
        for (int i = 0; i < W; ++i) mx = fmaxf(mx, v[r][q][i]);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
    float se = 0.f, xl = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int ch = q * 32 + lane;
      if (ch < nchunk) {
        const int rel = lab32 - ch * W;
#pragma unroll
        for (int i = 0; i < W; ++i) {
          const float e = __expf(v[r][q][i] - mx);
          se += e;
          if (i == rel) xl = v[r][q][i];
          v[r][q][i] = e;
        }
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { se += __shfl_xor(se, o, 32); xl += __shfl_xor(xl, o, 32); }
    const float lse = mx + __logf(se);
    if (loss_rows && lane == 0) loss_rows[row] = ignored ? 0.f : (lse - xl);
    if (!dlogits) continue;   // loss only (the gradient is produced by a second call in the backward pass)
    // row_scale: per-row upstream gradient of a reduction='none' loss (balanced category sampling: mask / N), times *scale_ptr
    const float scale = ignored ? 0.f : (row_scale ? scale0 * row_scale[row] : scale0);
    const float inv = scale / se;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int ch = q * 32 + lane;
      if (ch < nchunk) {
        const int rel = lab32 - ch * W;
#pragma unroll
        for (int i = 0; i < W; ++i) {
          float g = v[r][q][i] * inv;
          if (i == rel) g -= scale;
          v[r][q][i] = g;
        }
        if (vec) {
          LVec<T>::store(dlogits + row * c + ch * W, v[r][q]);
        } else {
#pragma unroll
          for (int i = 0; i < W; ++i)
            if (ch * W + i < c) st_elem(dlogits + row * c + ch * W + i, v[r][q][i]);
        }
      }
    }
  }
}

// rows the mean cross-entropy counts: integer atomics, one per workgroup -> deterministic
__global__ __launch_bounds__(256) void k_ce_count_valid(const int64_t *__restrict__ labels, int64_t n, int c, int64_t ignore_index,
                                                         int32_t *__restrict__ count) {
  __shared__ int32_t l_cnt;
  if (threadIdx.x == 0) l_cnt = 0;
  __syncthreads();
  int32_t mine = 0;
  const int64_t base = (int64_t)blockIdx.x * 256 * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t r = base + (int64_t)i * 256 + threadIdx.x;
    if (r < n) {
      const int64_t lab = labels[r];
      mine += (lab != ignore_index && lab >= 0 && lab < c) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&l_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0 && l_cnt) atomicAdd(count, l_cnt);
}

// head / common / tail statistics of the per-point losses (lib/losses/utils.py:69-72: loss[loss_items[:, g]] for the trainer's three
// meters, pl_BaselineTrainer.py:353-355) without the boolean-index gathers: per workgroup the sums and counts of the three groups
// -> partial[block][6]; the caller adds the <= 1024 rows up (deterministic, no float atomics).  One streaming pass: 12 B per point.
__global__ __launch_bounds__(256) void k_split_stats(const float *__restrict__ loss, const int64_t *__restrict__ labels, int64_t n,
                                                     const int32_t *__restrict__ group_of_class, int n_classes, int64_t ignore_index,
                                                     float *__restrict__ partial) {
  __shared__ float l_acc[4][6];
  float sum[3] = {0.f, 0.f, 0.f}, cnt[3] = {0.f, 0.f, 0.f};
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
    const int64_t lab = labels[r];
    if (lab == ignore_index || lab < 0 || lab >= n_classes) continue;
    const int g = group_of_class[lab];
    const float v = loss[r];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      sum[k] += g == k ? v : 0.f;
      cnt[k] += g == k ? 1.f : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sum[k] += __shfl_down(sum[k], o, 64);
      cnt[k] += __shfl_down(cnt[k], o, 64);
    }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { l_acc[wave][2 * k] = sum[k]; l_acc[wave][2 * k + 1] = cnt[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6)
    partial[(int64_t)blockIdx.x * 6 + threadIdx.x] = l_acc[0][threadIdx.x] + l_acc[1][threadIdx.x] + l_acc[2][threadIdx.x] + l_acc[3][threadIdx.x];
}

}  // namespace lgs

using namespace lgs;

extern "C" int lgs_split_stats(const float *loss_rows, const int64_t *labels, int64_t n, const int32_t *group_of_class, int n_classes,
                               int64_t ignore_index, float *partial, int partial_rows, void *stream) {
  LGS_REQUIRE(partial && group_of_class && n >= 0 && n_classes >= 1 && partial_rows >= 1 && partial_rows <= 1024 &&
                  ((loss_rows && labels) || n == 0),
              "lgs_split_stats: bad argument");
  LGS_KLAUNCH(k_split_stats, (unsigned)partial_rows, 256, 0, (hipStream_t)stream, loss_rows, labels, n, group_of_class, n_classes,
              ignore_index, partial);
  LGS_HIP(hipGetLastError());
  return 0;
}

extern "C" int lgs_ce_forward_backward_rows(const void *logits, int64_t n, int c, const int64_t *labels, int64_t ignore_index,
                                            const float *scale, const float *row_scale, float *loss_rows, void *dlogits, int dtype,
                                            void *stream) {
  LGS_REQUIRE(logits && labels && scale && (loss_rows || dlogits), "lgs_ce_forward_backward: null argument");
  const int W = dtype == LGS_BF16 ? 8 : 4;
  LGS_REQUIRE(c >= 1 && (c + W - 1) / W <= 32 * kMaxChunks, "lgs_ce_forward_backward: more classes than one half-wave holds (512 fp32 / 1024 bf16)");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (c + W - 1) / W, q = (nchunk + 31) / 32;
  // 8 half-waves per workgroup, R rows per half-wave (Q x R = 4)
#define LGS_CE_LAUNCH(T_, Q_, R_)                                                                                              \
  LGS_KLAUNCH((k_ce_fwd_bwd<T_, Q_, R_>), (unsigned)((n + 8 * (R_) - 1) / (8 * (R_))), 256, 0, s, (const T_ *)logits, n, c, labels, \
              ignore_index, scale, row_scale, loss_rows, (T_ *)dlogits)
  if (dtype == LGS_F32) {
    if (q <= 1) LGS_CE_LAUNCH(float, 1, 4); else if (q == 2) LGS_CE_LAUNCH(float, 2, 2); else LGS_CE_LAUNCH(float, 4, 1);
  } else if (dtype == LGS_BF16) {
    if (q <= 1) LGS_CE_LAUNCH(bf16_t, 1, 4); else if (q == 2) LGS_CE_LAUNCH(bf16_t, 2, 2); else LGS_CE_LAUNCH(bf16_t, 4, 1);
  } else {
    LGS_REQUIRE(false, "lgs_ce_forward_backward: unknown dtype");
  }
#undef LGS_CE_LAUNCH
  LGS_HIP(hipGetLastError());
  return 0;
}

int lgs_ce_forward_backward(const void *logits, int64_t n, int c, const int64_t *labels, int64_t ignore_index,
                            const float *scale, float *loss_rows, void *dlogits, int dtype, void *stream) {
  return lgs_ce_forward_backward_rows(logits, n, c, labels, ignore_index, scale, nullptr, loss_rows, dlogits, dtype, stream);
}

int lgs_ce_count_valid(const int64_t *labels, int64_t n, int c, int64_t ignore_index, int32_t *count, void *stream) {
  LGS_REQUIRE(count && (labels || n == 0) && n >= 0 && c >= 1, "lgs_ce_count_valid: bad argument");
  hipStream_t s = (hipStream_t)stream;
  LGS_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
  if (n == 0) return 0;
  LGS_KLAUNCH(k_ce_count_valid, (unsigned)((n + 2047) / 2048), 256, 0, s, labels, n, c, ignore_index, count);
  LGS_HIP(hipGetLastError());
  return 0;
}

// ---- backward of the fused CLIP text-anchor loss (lgs_clip_loss_forward)
//   /root/reference/lib/losses/ContrastiveLanguageLoss.py:73-95 (feat_dist, cos branch) through autograd:
// only 1 + K entries of a row of S = f^ . T^^T carry gradient, so instead of two dense [N, A] x [A, C] products
//   gf = ( sum_j gs_j t^_j  -  (sum_j gs_j s_j) f^ ) / |f|,      gs_pos = -g_dpos,  gs_neg_j = -g_dneg / K
// is a pure streaming pass: read f once, gather 1 + K normalised anchor rows (L2-resident table), write gf once.
// sum_j gs_j s_j needs only s_pos = 1 - d_pos and mean_j s_neg_j = 1 - d_neg, both saved by the forward.
namespace lgs {
template <typename T>
__global__ __launch_bounds__(256) void k_clip_loss_bwd(const T *__restrict__ feat, int64_t n, int c, const float *__restrict__ tn,
                                                       int n_anchor, const int64_t *__restrict__ labels,
                                                       const int64_t *__restrict__ neg, int k_neg, int64_t ignore,
                                                       const float *__restrict__ inv_norm, const float *__restrict__ d_pos,
                                                       const float *__restrict__ d_neg, const float *__restrict__ g_dpos,
                                                       const float *__restrict__ g_dneg, T *__restrict__ gf) {
  constexpr int W = LVec<T>::W;
  const int G = c / W;                       // 16-byte channel groups per row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = idx / G;
  const int g = (int)(idx - row * G);
  if (row >= n) return;
  float fv[W], out[W];
  LVec<T>::load(feat + row * c + g * W, fv);
  const int64_t lab = labels[row];
  const bool valid = lab != ignore && lab >= 0 && lab < n_anchor;
#pragma unroll
  for (int i = 0; i < W; ++i) out[i] = 0.f;
  if (valid) {
    const float inv = inv_norm[row];
    const float gp = g_dpos ? -g_dpos[row] : 0.f;
    const float gn = g_dneg ? -g_dneg[row] / (float)k_neg : 0.f;
    const float sigma = gp * (1.f - d_pos[row]) + gn * (float)k_neg * (1.f - d_neg[row]);
    const float *tp = tn + lab * c + g * W;
#pragma unroll
    for (int i = 0; i < W; i += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(tp + i);
      out[i] = gp * t.x; out[i + 1] = gp * t.y; out[i + 2] = gp * t.z; out[i + 3] = gp * t.w;
    }
    for (int j = 0; j < k_neg; ++j) {
      const int64_t a = neg[row * k_neg + j];
      if (a < 0 || a >= n_anchor) continue;
      const float *tq = tn + a * c + g * W;
#pragma unroll
      for (int i = 0; i < W; i += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(tq + i);
        out[i] += gn * t.x; out[i + 1] += gn * t.y; out[i + 2] += gn * t.z; out[i + 3] += gn * t.w;
      }
    }
    const float si = sigma * inv;
#pragma unroll
    for (int i = 0; i < W; ++i) out[i] = (out[i] - si * fv[i]) * inv;
  }
  LVec<T>::store(gf + row * c + g * W, out);
}
}  // namespace lgs

extern "C" int lgs_clip_loss_backward(const void *feat, int64_t n, int c, const float *anchors_n, int n_anchor,
                                      const int64_t *labels, const int64_t *neg, int k_neg, int64_t ignore_label,
                                      const float *inv_norm_f, const float *d_pos, const float *d_neg, const float *g_dpos,
                                      const float *g_dneg, void *grad_feat, int dtype, void *stream) {
  LGS_REQUIRE(feat && anchors_n && labels && neg && inv_norm_f && d_pos && d_neg && grad_feat, "lgs_clip_loss_backward: null argument");
  const int W = dtype == LGS_BF16 ? 8 : 4;
  LGS_REQUIRE(c % W == 0 && k_neg >= 1, "lgs_clip_loss_backward: feature dim must be a multiple of the 16-byte load width");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int64_t threads = n * (int64_t)(c / W);
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  if (dtype == LGS_F32)
    LGS_KLAUNCH((k_clip_loss_bwd<float>), blocks, 256, 0, s, (const float *)feat, n, c, anchors_n, n_anchor, labels, neg, k_neg,
                       ignore_label, inv_norm_f, d_pos, d_neg, g_dpos, g_dneg, (float *)grad_feat);
  else if (dtype == LGS_BF16)
    LGS_KLAUNCH((k_clip_loss_bwd<bf16_t>), blocks, 256, 0, s, (const bf16_t *)feat, n, c, anchors_n, n_anchor, labels, neg,
                       k_neg, ignore_label, inv_norm_f, d_pos, d_neg, g_dpos, g_dneg, (bf16_t *)grad_feat);
  else
    LGS_REQUIRE(false, "lgs_clip_loss_backward: unknown dtype");
  LGS_HIP(hipGetLastError());
  return 0;
}

// ---- fused SGD step on a flat bucket (torch.optim.SGD's rule, /root/reference/lib/solvers.py: momentum 0.9,
// dampening 0.1, weight decay 1e-4): d = g + wd * p;  buf = first ? d : m * buf + (1 - damp) * d;  p -= lr * mask * buf
// one pass over the bucket instead of four elementwise kernels
namespace lgs {
__global__ void k_sgd_step(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf,
                           const float *__restrict__ mask, int64_t n, float lr, float momentum, float dampening, float wd,
                           int first) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 pv = *reinterpret_cast<float4 *>(p + i);
    const float4 gv = *reinterpret_cast<const float4 *>(g + i);
    float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4 *>(buf + i);
    float4 mv = mask ? *reinterpret_cast<const float4 *>(mask + i) : make_float4(1.f, 1.f, 1.f, 1.f);
    float d;
#define LGS_SGD1(c)                                                                  \
    d = gv.c + wd * pv.c;                                                             \
    bv.c = momentum != 0.f ? (first ? d : momentum * bv.c + (1.f - dampening) * d) : d; \
    pv.c -= lr * mv.c * bv.c;
    LGS_SGD1(x) LGS_SGD1(y) LGS_SGD1(z) LGS_SGD1(w)
#undef LGS_SGD1
    if (momentum != 0.f) *reinterpret_cast<float4 *>(buf + i) = bv;
    *reinterpret_cast<float4 *>(p + i) = pv;
  } else {
    for (int64_t j = i; j < n; ++j) {
      const float d = g[j] + wd * p[j];
      float b = momentum != 0.f ? (first ? d : momentum * buf[j] + (1.f - dampening) * d) : d;
      if (momentum != 0.f) buf[j] = b;
      p[j] -= lr * (mask ? mask[j] : 1.f) * b;
    }
  }
}
}  // namespace lgs

extern "C" int lgs_sgd_step(float *params, const float *grads, float *momentum_buf, const float *mask, int64_t n, float lr,
                            float momentum, float dampening, float weight_decay, int first_step, void *stream) {
  LGS_REQUIRE(n >= 0 && (n == 0 || (params && grads && (momentum == 0.f || momentum_buf))), "lgs_sgd_step: bad argument");
  LGS_REQUIRE(((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(momentum_buf) |
                reinterpret_cast<uintptr_t>(mask)) & 15u) == 0, "lgs_sgd_step: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  const int64_t threads = (n + 3) / 4;
  LGS_KLAUNCH(lgs::k_sgd_step, (unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream, params, grads, momentum_buf,
                     mask, n, lr, momentum, dampening, weight_decay, first_step);
  LGS_HIP(hipGetLastError());
  return 0;
}
