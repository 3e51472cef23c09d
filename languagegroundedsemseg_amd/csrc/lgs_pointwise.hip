// lgs_pointwise.hip -- 1x1 stride-1 sparse convolution (= a dense [N, Cg] x [Cg, Co] product on the identity map) for the big
// maps, bf16, gfx950.
//
// Call sites (MinkowskiConvolution with kernel_size = 1, /root/reference/models/modules/common.py:195-203):
//   /root/reference/models/res16unet.py:193     `final`: 96 -> 200 classes + bias on every level-0 voxel
//   /root/reference/models/resnet.py:93-103     the 1x1 downsample branch of a stage's first BasicBlock (128 -> 96 at level 0)
// and their dgrads (200 -> 96, 96 -> 128).  On 1.2 M voxels these are pure streaming problems -- 96 -> 200 reads 231 MB and
// writes 482 MB for 46 GFLOP -- and k_conv_gather, built for gathered 3^3 tiles, ran them at 1.8 - 2.0 TB/s: one workgroup per
// 128 positions, each staging the whole weight matrix into LDS first (9418 workgroups x 43 KB), 234 VGPRs + 112 accumulation
// registers = one wave per SIMD, and load -> multiply -> store strictly one after the other inside it (profiles/r06_experiments.txt: 0.396 ms; this kernel 0.216).
//
// This kernel is the same MFMA arithmetic (v_mfma_f32_32x32x16_bf16, weights as the A operand so that a lane ends up owning 4
// consecutive output channels of ONE voxel, fp32 accumulation, one rounding to bf16) laid out for streaming:
//   * PERSISTENT workgroups (two per CU): the weight matrix is converted fp32 -> bf16 and laid out in MFMA-fragment order in
//     LDS ONCE per workgroup, then its four waves walk over 32-row blocks of the feature matrix;
//   * a wave's operand fragments ARE 16-byte row pieces (lane = (voxel, half): channels 16 ks + 8 half ...), loaded straight
//     from global memory into registers; the fragments of the NEXT row block are in flight while the current one is multiplied
//     and stored (two register sets);
//   * nothing else: no kernel map (identity), no index loads, no barrier after the prologue.
// Bytes: every input byte read once, every output byte written once, weights once per workgroup (512 x <= 60 KB).
#include "lgs_common.h"

namespace lgs {

typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));

// NB = output blocks of 32 channels per row block (all of them: one pass over the input), KS = 16-channel reduction steps held
// in registers (x 2 sets).  Register budget per lane: 16 NB accumulators + 8 KS fragment registers + ~30.
template <int NB, int KS, int OCC>
__global__ __launch_bounds__(256, OCC) void k_pointwise(const bf16_t *__restrict__ in, int64_t in_ld, int g_real,
                                                      const float *__restrict__ w, int cin_w, int cout_w, int transposed,
                                                      int o_real, const float *__restrict__ bias, bf16_t *__restrict__ out,
                                                      int64_t n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4p *l_w = reinterpret_cast<u32x4p *>(smem);                      // [nks][NB][64] weight fragments
  const int nks = (g_real + 15) / 16;
  float *l_bias = reinterpret_cast<float *>(smem + (size_t)nks * NB * 64 * 16);   // [NB * 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 31, h = lane >> 5;

  // ---- prologue: weights fp32 [cin_w, cout_w] -> bf16 fragments.  Fragment (ks, nb), lane (j, h): output channel o = 32 nb + j,
  // reduction elements g = 16 ks + 8 h + e, e = 0..7; transposed (dgrad): the reduction runs over the weight's second index
  for (int idx = tid; idx < nks * NB * 64; idx += 256) {
    const int ln = idx & 63, nb = (idx >> 6) % NB, ks = (idx >> 6) / NB;
    const int o = nb * 32 + (ln & 31), g0 = ks * 16 + (ln >> 5) * 8;
    uint32_t pk[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float x[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int g = g0 + 2 * e2 + t;
        x[t] = (g < g_real && o < o_real) ? (transposed ? w[(int64_t)o * cout_w + g] : w[(int64_t)g * cout_w + o]) : 0.f;
      }
      pk[e2] = (uint32_t)f32_to_bf16(x[0]) | ((uint32_t)f32_to_bf16(x[1]) << 16);
    }
    l_w[idx] = u32x4p{pk[0], pk[1], pk[2], pk[3]};
  }
  for (int c = tid; c < NB * 32; c += 256) l_bias[c] = (bias && c < o_real) ? bias[c] : 0.f;
  __syncthreads();

  const int64_t nrb = (n + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t rb = (int64_t)blockIdx.x * 4 + wave;
  if (rb >= nrb) return;

  u32x4p cur[KS], nxt[KS];
  auto fetch = [&](int64_t blk, u32x4p (&f)[KS]) __attribute__((always_inline)) {
    const int64_t row = blk * 32 + vx;
    const bf16_t *p = in + row * in_ld + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f[ks] = u32x4p{0u, 0u, 0u, 0u};
      if (ks < nks && row < n && ks * 16 + h * 8 < g_real) f[ks] = *reinterpret_cast<const u32x4p *>(p + ks * 16);
    }
  };
  fetch(rb, cur);
  while (true) {
    const int64_t nrb_next = rb + stride;
    const bool more = nrb_next < nrb;
    if (more) fetch(nrb_next, nxt);                 // in flight under the multiply and the stores of this block
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < nks) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const u32x4p a = l_w[(ks * NB + nb) * 64 + lane];
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, cur[ks]), acc[nb], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: lane (voxel vx, half h) owns channels 32 nb + 8 q + 4 h + {0..3} of its row.  Rows on the 8-channel grid are
    // written with SIXTEEN-byte stores: v_permlane32_swap trades piece q = 2p + 1 of the h = 0 lane for piece q = 2p of the h = 1
    // lane, so that lane h owns the 8 contiguous channels 32 nb + 16 p + 8 h .. + 7 (the idiom of k_conv_gather's epilogue)
    const int64_t row = rb * 32 + vx;
    bf16_t *dst = out + (row < n ? row : 0) * o_real;
    if ((o_real & 7) == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int pq = 0; pq < 2; ++pq) {
          uint32_t pk[2][2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * pq + e, c0 = nb * 32 + 8 * q + 4 * h;
            const float4 b = *reinterpret_cast<const float4 *>(l_bias + c0);
            pk[e][0] = (uint32_t)f32_to_bf16(acc[nb][4 * q + 0] + b.x) | ((uint32_t)f32_to_bf16(acc[nb][4 * q + 1] + b.y) << 16);
            pk[e][1] = (uint32_t)f32_to_bf16(acc[nb][4 * q + 2] + b.z) | ((uint32_t)f32_to_bf16(acc[nb][4 * q + 3] + b.w) << 16);
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
          const int c8 = nb * 32 + 16 * pq + 8 * h;
          if (row < n && c8 < o_real) *reinterpret_cast<u32x4p *>(dst + c8) = u32x4p{s0[0], s1[0], s0[1], s1[1]};
        }
      }
    } else if (row < n) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = nb * 32 + 8 * q + 4 * h;
          if (c0 < o_real) {
            const float4 b = *reinterpret_cast<const float4 *>(l_bias + c0);
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(acc[nb][4 * q + 0] + b.x) | ((uint32_t)f32_to_bf16(acc[nb][4 * q + 1] + b.y) << 16);
            pk.y = (uint32_t)f32_to_bf16(acc[nb][4 * q + 2] + b.z) | ((uint32_t)f32_to_bf16(acc[nb][4 * q + 3] + b.w) << 16);
            *reinterpret_cast<uint2 *>(dst + c0) = pk;
          }
        }
      }
    }
    if (!more) break;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) cur[ks] = nxt[ks];
    rb = nrb_next;
  }
}

// ---- fp32 (the parity path).  Same structure on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: every product and every sum in fp32,
// the arithmetic of FP32_SPLIT=0), which at 64 cycles per instruction stays under the memory time of these layers.  The identity-map
// tiles of k_conv_gather ran them far off the bytes: 96 -> 200 forward 0.80 ms, 200 -> 96 dgrad 1.06, 96 -> 128 dgrad 2.3 ms
// (profiles/r05_kernel_stats_fp32.txt) for 1.1 - 1.4 GB each; rocBLAS needs 0.34 - 0.72 ms (tools/dbg/gemm_ref.py).
// A lane's operand piece is a float4 (channels 8 j + 4 h + e): element e of the lanes h = 0 / 1 is the k-pair of MFMA step (j, e).
// The reduction runs in chunks of 64 channels (eight float4 per lane, two register sets): the next chunk -- of this row block or
// the first of the next -- is in flight under the current chunk's 32 NB MFMAs.
template <int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void k_pointwise_f32(const float *__restrict__ in, int64_t in_ld, int g_real,
                                                            const float *__restrict__ w, int cin_w, int cout_w, int transposed,
                                                            int o_real, const float *__restrict__ bias, float *__restrict__ out,
                                                            int64_t n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *l_w = reinterpret_cast<float4 *>(smem);                      // [nj][NB][64]: W[8 j + 4 h + e][32 nb + o], e = 0..3
  const int nj = (g_real + 7) / 8, nchunk = (nj + 7) / 8;
  float *l_bias = reinterpret_cast<float *>(smem + (size_t)nj * NB * 64 * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 31, h = lane >> 5;
  for (int idx = tid; idx < nj * NB * 64; idx += 256) {
    const int ln = idx & 63, nb = (idx >> 6) % NB, j = (idx >> 6) / NB;
    const int o = nb * 32 + (ln & 31), g0 = j * 8 + (ln >> 5) * 4;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = g0 + e;
      x[e] = (g < g_real && o < o_real) ? (transposed ? w[(int64_t)o * cout_w + g] : w[(int64_t)g * cout_w + o]) : 0.f;
    }
    l_w[idx] = make_float4(x[0], x[1], x[2], x[3]);
  }
  for (int c = tid; c < NB * 32; c += 256) l_bias[c] = (bias && c < o_real) ? bias[c] : 0.f;
  __syncthreads();

  const int64_t nrb = (n + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t rb = (int64_t)blockIdx.x * 4 + wave;
  if (rb >= nrb) return;

  float4 cur[8], nxt[8];
  auto fetch = [&](int64_t blk, int chunk, float4 (&f)[8]) __attribute__((always_inline)) {
    const int64_t row = blk * 32 + vx;
    const float *p = in + row * in_ld + h * 4;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = chunk * 8 + jj;
      f[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < nj && row < n && j * 8 + h * 4 < g_real) f[jj] = *reinterpret_cast<const float4 *>(p + j * 8);
    }
  };
  fetch(rb, 0, cur);
  int chunk = 0;
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  while (true) {
    // what comes after this chunk: the next chunk of the row block, or the first chunk of this wave's next row block
    const bool last_chunk = chunk + 1 >= nchunk;
    const int64_t rb_n = last_chunk ? rb + stride : rb;
    const int chunk_n = last_chunk ? 0 : chunk + 1;
    const bool more = rb_n < nrb;
    if (more) fetch(rb_n, chunk_n, nxt);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = chunk * 8 + jj;
      if (j < nj) {
        const float c4[4] = {cur[jj].x, cur[jj].y, cur[jj].z, cur[jj].w};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float4 a = l_w[(j * NB + nb) * 64 + lane];
          const float a4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], c4[e], acc[nb], 0, 0, 0);
        }
      }
    }
    if (last_chunk) {
      // ---- epilogue: lane (voxel vx, half h) owns channels 32 nb + 8 q + 4 h + {0..3} of its row: one 16-byte store each
      const int64_t row = rb * 32 + vx;
      if (row < n) {
        float *dst = out + row * o_real;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = nb * 32 + 8 * q + 4 * h;
            if (c0 < o_real) {
              const float4 b = *reinterpret_cast<const float4 *>(l_bias + c0);
              *reinterpret_cast<float4 *>(dst + c0) = make_float4(acc[nb][4 * q + 0] + b.x, acc[nb][4 * q + 1] + b.y,
                                                                  acc[nb][4 * q + 2] + b.z, acc[nb][4 * q + 3] + b.w);
            }
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    }
    if (!more) break;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) cur[jj] = nxt[jj];
    rb = rb_n;
    chunk = chunk_n;
  }
}

constexpr size_t kPwF32MaxLds = 144 * 1024;     // (128 -> 192, the level-2 downsample branch's dgrad: 16 x 7 KB of fragments, one workgroup per CU)

bool pointwise_f32_supported(const View &v, int K, int g_real, int o_real, int64_t in_ld) {
  if (tune(T_POINTWISE) == 0) return false;
  if (K != 1 || v.KS != 1 || v.nbr || v.out_row || v.tile_k || v.n_in != v.n_out || v.n_out < 65536) return false;
  if (g_real % 4 != 0 || o_real % 4 != 0 || (in_ld * 4) % 16 != 0) return false;
  const int nb = (o_real + 31) / 32, nj = (g_real + 7) / 8;
  if (nb < 3 || nb > 7) return false;
  const int nbt = nb >= 5 ? 7 : nb;
  return (size_t)nj * nbt * 64 * 16 + (size_t)nbt * 32 * 4 <= kPwF32MaxLds;
}

int launch_pointwise_f32(const View &v, const void *in, int64_t in_ld, int g_real, const float *w, int cin_w, int cout_w, int transposed,
                         int o_real, const float *bias, void *out, hipStream_t s) {
  const int64_t n = v.n_out;
  if (n == 0) return 0;
  LGS_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0,
              "pointwise conv: feature rows must start 16-byte aligned");
  const int nb = (o_real + 31) / 32, nj = (g_real + 7) / 8;
  const int nbt = nb >= 5 ? 7 : nb;
  const size_t lds = (size_t)nj * nbt * 64 * 16 + (size_t)nbt * 32 * 4;
  LGS_REQUIRE(lds <= kPwF32MaxLds, "pointwise conv (fp32): weight fragments exceed the LDS budget (internal error)");
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  const int64_t nrb = (n + 31) / 32;
  const int occ = lds * 2 <= 156 * 1024 ? 2 : 1;            // workgroups per CU the LDS holds (the register budget allows two)
  const float *fi = reinterpret_cast<const float *>(in);
  float *fo = reinterpret_cast<float *>(out);
#define LGS_PWF_LAUNCH(NB_)                                                                                                       \
  do {                                                                                                                            \
    static bool attr = false;                                                                                                     \
    if (!attr) {                                                                                                                  \
      LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pointwise_f32<NB_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwF32MaxLds)); \
      attr = true;                                                                                                                \
    }                                                                                                                             \
    unsigned grid = (unsigned)(occ * n_cu);                                                                                       \
    if ((int64_t)grid * 4 > nrb) grid = (unsigned)((nrb + 3) / 4);                                                                \
    LGS_KLAUNCH((k_pointwise_f32<NB_, 2>), grid, 256, lds, s, fi, in_ld, g_real, w, cin_w, cout_w, transposed, o_real, bias, fo, n); \
  } while (0)
  if (nbt == 7) LGS_PWF_LAUNCH(7);
  else if (nbt == 4) LGS_PWF_LAUNCH(4);
  else LGS_PWF_LAUNCH(3);
#undef LGS_PWF_LAUNCH
  LGS_HIP(hipGetLastError());
  return 0;
}

// -> true if this launch shape is served (the caller falls back to k_conv_gather otherwise)
bool pointwise_supported(const View &v, int K, int g_real, int o_real, int64_t in_ld) {
  if (tune(T_POINTWISE) == 0) return false;
  if (K != 1 || v.KS != 1 || v.nbr || v.out_row || v.tile_k || v.n_in != v.n_out || v.n_out < 65536) return false;
  if (g_real % 8 != 0 || o_real % 4 != 0 || (in_ld * 2) % 16 != 0) return false;
  const int nb = (o_real + 31) / 32, nks = (g_real + 15) / 16;
  // where it wins (tools/dbg/pointwise_ab.py, 1.2 M rows): 5 - 7 output blocks (96 -> 200 forward: 0.216 against 0.396 ms) and long
  // reductions into 3 blocks (its dgrad 200 -> 96: 0.187 against 0.215).  The narrow shapes (128 -> 96, 96 -> 96, 96 -> 128) are as
  // fast or faster on k_conv_gather's 3-waves-per-SIMD tiles (0.135 against 0.147 ms): they stay there unless POINTWISE=2
  const bool all = tune(T_POINTWISE) == 2;
  if (nb >= 5 && nb <= 7) return nks <= 8;
  if (nb == 4) return all && nks <= 8;
  if (nb == 3) return nks <= 16 && (all || nks > 8);
  return false;
}

int launch_pointwise(const View &v, const void *in, int64_t in_ld, int g_real, const float *w, int cin_w, int cout_w, int transposed,
                     int o_real, const float *bias, void *out, hipStream_t s) {
  const int64_t n = v.n_out;
  if (n == 0) return 0;
  LGS_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0,
              "pointwise conv: feature rows must start 16-byte aligned");
  const int nb = (o_real + 31) / 32, nks = (g_real + 15) / 16;
  const int nbt = nb >= 5 ? 7 : nb;                      // blocks of the kernel instance that serves it
  const size_t lds = (size_t)nks * nbt * 64 * 16 + (size_t)nbt * 32 * 4;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  const int64_t nrb = (n + 31) / 32;
  const bf16_t *fi = reinterpret_cast<const bf16_t *>(in);
  bf16_t *fo = reinterpret_cast<bf16_t *>(out);
#define LGS_PW_LAUNCH(NB_, KS_, OCC_)                                                                                             \
  do {                                                                                                                            \
    static bool attr = false;                                                                                                     \
    if (!attr) {                                                                                                                  \
      LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pointwise<NB_, KS_, OCC_>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024)); \
      attr = true;                                                                                                                \
    }                                                                                                                             \
    unsigned grid = (unsigned)((OCC_) * n_cu);                                                                                    \
    if ((int64_t)grid * 4 > nrb) grid = (unsigned)((nrb + 3) / 4);                                                                \
    LGS_KLAUNCH((k_pointwise<NB_, KS_, OCC_>), grid, 256, lds, s, fi, in_ld, g_real, w, cin_w, cout_w, transposed, o_real, bias, fo, n); \
  } while (0)
  LGS_REQUIRE(lds <= 72 * 1024, "pointwise conv: weight fragments exceed the LDS budget (internal error)");
  if (nb >= 5 && nb <= 7) LGS_PW_LAUNCH(7, 8, 2);
  else if (nb == 4) LGS_PW_LAUNCH(4, 8, 3);
  else if (nb == 3 && nks <= 8) LGS_PW_LAUNCH(3, 8, 3);
  else if (nb == 3) LGS_PW_LAUNCH(3, 16, 2);
  else LGS_REQUIRE(false, "pointwise conv: unsupported shape (internal error)");
#undef LGS_PW_LAUNCH
  LGS_HIP(hipGetLastError());
  return 0;
}

}  // namespace lgs
