// lgs_conv.hip -- sparse convolution on gfx950: forward, dgrad, wgrad.
//
// Replaces the arithmetic of MinkowskiConvolution / MinkowskiConvolutionTranspose as called from
//   /root/reference/models/modules/common.py:195-203,228-236  (conv(), conv_tr())
//   /root/reference/models/modules/resnet_block.py:41-57       (BasicBlock.forward)
//   /root/reference/models/res16unet.py:196-270                 (Res16UNetBase.forward)
//
// Formulation: OUTPUT-STATIONARY IMPLICIT GEMM.  A workgroup owns a tile of
// Morton-consecutive output positions and a tile of output channels; for every kernel offset that has
// any neighbour in the tile (wavefront-ballot bitmask built with the kernel map) it gathers the
// neighbour rows straight from HBM into MFMA operand registers (16 B per lane, whole 32/64-byte row
// pieces), multiplies by the offset's weight slice staged once per workgroup in LDS (pre-packed in
// MFMA fragment order so the copy is linear and ds_read_b128 is conflict-free), and accumulates in
// fp32.  Every output row is written exactly once: no atomics, deterministic.  dgrad is the same
// kernel on transposed (and, for 3x3x3, mirrored) weights; transposed convs use the grouped view.
//
// MFMA use: v_mfma_f32_32x32x16_bf16 for bf16 storage, v_mfma_f32_32x32x2_f32 (exact fp32) for fp32.
// Operands are swapped (weights = A, voxels = B) so a lane ends up owning 4 consecutive output
// channels of one voxel -> 8/16-byte stores.  The reduction index (channel) is permuted identically on
// both operands so every lane's load is 16 contiguous bytes.
#include "lgs_common.h"

#include <type_traits>

#include <stdlib.h>
#include <string.h>

namespace lgs {


// fp32 STORAGE whose products run on the bf16 matrix pipe (round 5): every fp32 operand is split exactly into three bf16 pieces
// x = hi + mid + lo (8 + 8 + 8 significant bits, by truncation: each remainder is exact), and x * w is accumulated in fp32 from
// the six products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- what is dropped (mid*lo, lo*mid, lo*lo) is below 2^-24 |x w|.
// v_mfma_f32_32x32x16_bf16 runs 16 x the flops per cycle of v_mfma_f32_32x32x2_f32: six products are 2.7 x faster on the pipe.
// The feature operand is split in registers as it streams through (11 VALU instructions per pair of elements, reused by every
// output-channel block of the wave); the weights are split once, when the packed image is made (three pieces per element).
struct f32s_t { float v; };
constexpr int kDtF32Split = 2;      // lgs_pack_desc.dtype of an image packed for f32s_t (internal; tensors are LGS_F32)
template <typename T> struct Tr;
template <> struct Tr<float> {
  static constexpr int EPL = 4;  // elements per 16-byte load
  static constexpr int LD = 4;   // 16-byte loads per lane per 32-channel chunk
  static constexpr int WLD = 4;  // 16-byte weight fragments per lane, chunk and output block
  static constexpr bool SPLIT = false;
};
template <> struct Tr<bf16_t> {
  static constexpr int EPL = 8;
  static constexpr int LD = 2;
  static constexpr int WLD = 2;
  static constexpr bool SPLIT = false;
};
template <> struct Tr<f32s_t> {
  static constexpr int EPL = 4;
  static constexpr int LD = 4;
  static constexpr int WLD = 6;  // [k-step of 16 channels: 2][piece hi / mid / lo: 3]
  static constexpr bool SPLIT = true;
};


// one 16-byte operand pair: bf16 = one 32x32x16 MFMA, fp32 = four 32x32x2 MFMAs
// 16-byte operands are native ext-vectors (not the HIP_vector_type struct): struct copies between address spaces
// lower to llvm.memcpy, which keeps the staging array in scratch memory instead of registers.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ inline void mma16(f32x16 &acc, const u32x4 &w, const u32x4 &f);
template <> __device__ inline void mma16<bf16_t>(f32x16 &acc, const u32x4 &w, const u32x4 &f) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, f), acc, 0, 0, 0);
}
__device__ inline void mma_bf16(f32x16 &acc, const u32x4 &w, const u32x4 &f) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, f), acc, 0, 0, 0);
}
// one fp32 value -> its three bf16 pieces as the upper halves of three dwords (hi + mid + lo == x exactly)
__device__ inline void split3(uint32_t x, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  hi = x & 0xffff0000u;
  const float r1 = __uint_as_float(x) - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));          // <= 8 significant bits: its lower half is zero
}
// eight consecutive fp32 reduction elements (two 16-byte loads) -> three bf16x8 MFMA operands
__device__ inline void split8(const u32x4 &a, const u32x4 &b, u32x4 &hi, u32x4 &mid, u32x4 &lo) {
  const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split3(x[i], h[i], m[i], l[i]);
  // element 2i in the low half, 2i + 1 in the high half of dword i: v_perm_b32 picks the upper halves of both
#define LGS_PK(v, i) __builtin_amdgcn_perm(v[2 * (i) + 1], v[2 * (i)], 0x07060302u)
  hi = u32x4{LGS_PK(h, 0), LGS_PK(h, 1), LGS_PK(h, 2), LGS_PK(h, 3)};
  mid = u32x4{LGS_PK(m, 0), LGS_PK(m, 1), LGS_PK(m, 2), LGS_PK(m, 3)};
  lo = u32x4{LGS_PK(l, 0), LGS_PK(l, 1), LGS_PK(l, 2), LGS_PK(l, 3)};
#undef LGS_PK
}
template <> __device__ inline void mma16<f32s_t>(f32x16 &acc, const u32x4 &w, const u32x4 &f) { (void)acc; (void)w; (void)f; }   // (unused: see compute)
template <> __device__ inline void mma16<float>(f32x16 &acc, const u32x4 &w, const u32x4 &f) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(f.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(f.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(f.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(f.w), acc, 0, 0, 0);
}

// Missing neighbours / padded channels are not branched around (hipcc would fence every such load with
// s_waitcnt vmcnt(0) and serialise the whole gather pipeline): their lane gets an out-of-range buffer offset, for
// which the hardware returns zeros.

// ------------------------------------------------------------------------------------ weight packing
// dst[kd][c][nb][t][lane] (16 B each): element e of lane (j = lane&31, h = lane>>5) is
//   Wsrc[g = c*32 + h*16 + t*EPL + e][o = nb*32 + j]   of weight matrix kd
// where (g = gathered/reduction channel, o = output channel):
//   plain      : Wsrc[g][o] = w[ks][g][o],        ks = kd
//   transposed : Wsrc[g][o] = w[ks][o][g],        ks = mirror ? K-1-kd : kd      (dgrad)
// Out-of-range g / o are zero (channel padding to multiples of 32).
template <typename T>
__device__ inline uint4 pack_one(const float *__restrict__ w, int K, int cin_w, int cout_w, int transposed, int mirror, int g_real,
                                 int o_real, int nc, int nb_total, int64_t idx) {
  constexpr int EPL = Tr<T>::EPL, LD = Tr<T>::WLD;
  int lane = (int)(idx & 63);
  int64_t r = idx >> 6;
  int t = (int)(r % LD); r /= LD;
  int nb = (int)(r % nb_total); r /= nb_total;
  int c = (int)(r % nc);
  int kd = (int)(r / nc);
  int ks = (transposed && mirror) ? K - 1 - kd : kd;
  int j = lane & 31, h = lane >> 5;
  int o = nb * 32 + j;
  if constexpr (Tr<T>::SPLIT) {
    // fragment t = 3 tt + piece: the eight reduction elements g = c*32 + h*16 + tt*8 + e of k-step tt, as bf16 piece `piece`
    const int tt = t / 3, piece = t % 3;
    uint32_t pc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = c * 32 + h * 16 + tt * 8 + e;
      float x = 0.f;
      if (g < g_real && o < o_real)
        x = transposed ? w[((int64_t)ks * cin_w + o) * cout_w + g] : w[((int64_t)ks * cin_w + g) * cout_w + o];
      uint32_t hi, mid, lo;
      split3(__float_as_uint(x), hi, mid, lo);
      pc[e] = piece == 0 ? hi : (piece == 1 ? mid : lo);
    }
    return make_uint4((pc[0] >> 16) | pc[1], (pc[2] >> 16) | pc[3], (pc[4] >> 16) | pc[5], (pc[6] >> 16) | pc[7]);
  }
  float vals[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    int g = c * 32 + h * 16 + t * EPL + e;
    float x = 0.f;
    if (g < g_real && o < o_real)
      x = transposed ? w[((int64_t)ks * cin_w + o) * cout_w + g] : w[((int64_t)ks * cin_w + g) * cout_w + o];
    vals[e] = x;
  }
  uint4 out;
  if constexpr (EPL == 4) {
    out = make_uint4(__float_as_uint(vals[0]), __float_as_uint(vals[1]), __float_as_uint(vals[2]), __float_as_uint(vals[3]));
  } else {
    out.x = (uint32_t)f32_to_bf16(vals[0]) | ((uint32_t)f32_to_bf16(vals[1]) << 16);
    out.y = (uint32_t)f32_to_bf16(vals[2]) | ((uint32_t)f32_to_bf16(vals[3]) << 16);
    out.z = (uint32_t)f32_to_bf16(vals[4]) | ((uint32_t)f32_to_bf16(vals[5]) << 16);
    out.w = (uint32_t)f32_to_bf16(vals[6]) | ((uint32_t)f32_to_bf16(vals[7]) << 16);
  }
  return out;
}

template <typename T>
__global__ void k_pack_weights(const float *__restrict__ w, int K, int cin_w, int cout_w, int transposed, int mirror,
                               int g_real, int o_real, int nc /*padded chunks*/, int nb_total /*padded blocks*/,
                               uint4 *__restrict__ dst) {
  constexpr int LD = Tr<T>::WLD;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)K * nc * nb_total * LD * 64;
  if (idx >= total) return;
  dst[idx] = pack_one<T>(w, K, cin_w, cout_w, transposed, mirror, g_real, o_real, nc, nb_total, idx);
}

// every cached packed image of the model in ONE launch (after the optimiser step): blockIdx.y = descriptor
__global__ void k_pack_weights_batch(const lgs_pack_desc *__restrict__ descs) {
  const lgs_pack_desc e = descs[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= e.total) return;
  uint4 *dst = reinterpret_cast<uint4 *>(e.packed);
  if (e.dtype == LGS_BF16) dst[idx] = pack_one<bf16_t>(e.weight, e.K, e.cin_w, e.cout_w, e.transposed, e.mirror, e.g_real, e.o_real, e.ncp, e.nbp, idx);
  else if (e.dtype == kDtF32Split) dst[idx] = pack_one<f32s_t>(e.weight, e.K, e.cin_w, e.cout_w, e.transposed, e.mirror, e.g_real, e.o_real, e.ncp, e.nbp, idx);
  else dst[idx] = pack_one<float>(e.weight, e.K, e.cin_w, e.cout_w, e.transposed, e.mirror, e.g_real, e.o_real, e.ncp, e.nbp, idx);
}

// (descriptor dtype code of a split-fp32 image: internal to the packed-image descriptors, never a tensor dtype)
// pad rows [n, c] -> [n, cpad] (zero fill) for channel counts that are not a multiple of the load width
template <typename T>
__global__ void k_pad_rows(const T *__restrict__ src, int64_t n, int c, int cpad, T *__restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * cpad) return;
  int64_t r = i / cpad;
  int ch = (int)(i % cpad);
  dst[i] = ch < c ? src[r * c + ch] : (T)0;
}

// inverse of k_pad_rows: [n, cpad] -> [n, c]
template <typename T>
__global__ void k_unpad_rows(const T *__restrict__ src, int64_t n, int c, int cpad, T *__restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  dst[i] = src[(i / c) * cpad + (i % c)];
}

// ------------------------------------------------------------------------------------ CLIP-loss epilogue (EPI = 1)
// The contraction S = normalize(F) . normalize(T)^T of ContrastiveLanguageLoss.feat_dist / feature_sim
// (/root/reference/lib/losses/ContrastiveLanguageLoss.py:73-95,185-192, lib/losses/utils.py:80-103) never needs its
// [N, n_anchor] result in memory: the loss reads 1 + K entries per row and the metrics read the arg-max.  With EPI = 1
// the tile's accumulators are reduced in the epilogue to  d_pos = 1 - s[label],  d_neg = 1 - mean_j s[neg_j],
// pred = argmax_a s[a]  and  1/|f|  (|f|^2 is accumulated from the operand fragments as they stream through, so the
// features are read exactly once); the similarity matrix itself is written only when the caller asks for it.
struct ClipEpi {
  const int64_t *labels = nullptr;   // [n]
  const int64_t *neg = nullptr;      // [n, k_neg] negative anchor indices
  int k_neg = 0;                     // 1..7
  int n_anchor = 0;
  int64_t ignore = -1;
  float *d_pos = nullptr, *d_neg = nullptr, *inv_norm = nullptr;
  int64_t *pred = nullptr;
};
// BatchNorm statistics from the conv epilogue (EPI = 0): when `partial` is set, every workgroup also writes, for its
// rows and channels, sum(y - pivot) and sum((y - pivot)^2) of the values it STORES (after the bf16 rounding, so they
// are the statistics of the tensor BatchNorm will read) to partial[tile][2][cout]: the [N, C] output is not read again
// for the statistics pass (k_colreduce<0> was ~6 % of the step).  pivot (may be NULL = 0) is a per-channel shift that
// keeps var = E[d^2] - E[d]^2 well conditioned; the caller passes BatchNorm's running mean.
struct BnEpi {
  float *partial = nullptr;     // [gridDim.x][2][cout_real]
  const float *pivot = nullptr; // [cout_real] or NULL
  int accum = 0;                // 1: out += result (lgs_conv_dgrad_accumulate: the residual branch's gradient is already in `out`)
};
// four adjacent stored elements -> fp32 (one 8- or 16-byte access)
__device__ inline void load4(const float *p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ inline void load4(const bf16_t *p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2 *>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ inline void load4(const f32s_t *p, float (&v)[4]) { load4(reinterpret_cast<const float *>(p), v); }
template <typename T> __device__ inline float stored_value(float x);
template <> __device__ inline float stored_value<f32s_t>(float x) { return x; }
template <> __device__ inline float stored_value<float>(float x) { return x; }
template <> __device__ inline float stored_value<bf16_t>(float x) { return bf16_to_f32(f32_to_bf16(x)); }
template <typename T> __device__ inline float sq16(const u32x4 &f, float s);
template <> __device__ inline float sq16<bf16_t>(const u32x4 &f, float s) {
  // two bf16 per dword: the high one IS an fp32 with the low half masked off, the low one is a 16-bit shift away
  const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
    s = fmaf(hi, hi, fmaf(lo, lo, s));
  }
  return s;
}
template <> __device__ inline float sq16<f32s_t>(const u32x4 &f, float s) {
  const float a = __uint_as_float(f.x), b = __uint_as_float(f.y), c = __uint_as_float(f.z), d = __uint_as_float(f.w);
  return fmaf(d, d, fmaf(c, c, fmaf(b, b, fmaf(a, a, s))));
}
template <> __device__ inline float sq16<float>(const u32x4 &f, float s) {
  const float a = __uint_as_float(f.x), b = __uint_as_float(f.y), c = __uint_as_float(f.z), d = __uint_as_float(f.w);
  return fmaf(d, d, fmaf(c, c, fmaf(b, b, fmaf(a, a, s))));
}

// ------------------------------------------------------------------------------------ forward / dgrad
// Tile: WM x WN waves; each wave owns RB*32 positions x NCB*32 output channels.
template <typename T, int RB, int NCB, int WM, int WN, int SC, int D, int EPI = 0>
__global__ __launch_bounds__(WM *WN * 64) void k_conv_gather(View v, const T *__restrict__ in, int cin_real, int nc,
                                                             const u32x4 *__restrict__ wp, int nb_total,
                                                             int ncp, int nbp,
                                                             T *__restrict__ out, int cout_real,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ out_f32_arg,
                                                             const float *__restrict__ row_scale,
                                                             unsigned in_bytes, unsigned w_bytes, int64_t zstride,
                                                             ClipEpi ce, BnEpi be, int gc, int in_ld) {
  static_assert(EPI == 0 || (RB == 1 && WN == 1), "the CLIP epilogue owns whole rows: one row block, all columns per wave");
  // SC = 32-channel chunks per weight SLAB: the weights of (offset, slab) are staged in LDS once per workgroup
  // and one barrier separates slabs, while the gathered feature fragments stream chunk by chunk through a
  // D-deep register ring (loads issued D-1 chunks = several hundred MFMA cycles ahead of their use) that runs
  // across slab and offset boundaries.
  constexpr int EPL = Tr<T>::EPL, LD = Tr<T>::LD, WLD = Tr<T>::WLD;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = WM * RB * 32;
  constexpr int WB = WN * NCB;                 // weight blocks per chunk
  constexpr int WCH = WB * WLD * 64;           // uint4 per chunk
  constexpr int SLAB = SC * WCH;               // uint4 per slab
  constexpr int WR = (SLAB + NT - 1) / NT;     // staging registers per thread
  __shared__ u32x4 lds[2][SLAB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int vx = lane & 31, h = lane >> 5;
  // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (speed only, never correctness), so give
  // every XCD a CONTIGUOUS run of position tiles -- neighbouring tiles share most of their gathered rows and
  // now hit the same private L2.  Bijective for any tile count.
  int64_t tile;
  {
    const unsigned nt = gridDim.x, xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, q = nt >> 3, r = nt & 7u;
    tile = (int64_t)(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int64_t pos_wg = tile * TM;
  const int64_t pos_w = pos_wg + (int64_t)wm * RB * 32;
  // Eight-wave tiles (one 32-row block per wave) INTERLEAVE the tile's rows over the waves (row i -> wave i % 8): the
  // rows are sorted by neighbourhood shape, so contiguous blocks would leave whole waves without work at an offset while
  // the others run their MFMAs, and everybody meets at the next slab barrier.
  constexpr bool ILV = (WM == 8 && RB == 1 && EPI == 0);
  auto row_of = [&](int rb) __attribute__((always_inline)) { return ILV ? vx * WM + wm : wm * RB * 32 + rb * 32 + vx; };
  const int nb_wg = blockIdx.y * WB;  // first cout block of the workgroup
  const int nb_w = nb_wg + wn * NCB;  // first cout block of this wave

  // ---- which slots does this workgroup visit, with which weight matrix
  uint32_t smask = 1;  // single slot
  int kw_single = 0;
  if (v.KS > 1) {
    smask = 0;
#pragma unroll
    for (int g = 0; g < (TM + 63) / 64; ++g) smask |= v.mask64[pos_wg / 64 + g];
  } else if (v.tile_k) {
    kw_single = v.tile_k[pos_wg / 64];
    if (kw_single < 0) {        // padding group: nothing to write (its statistics row is all zeros)
      if constexpr (EPI == 0) {
        if (be.partial != nullptr)
          for (int e = tid; e < 2 * WB * 32; e += NT) {
            const int st = e / (WB * 32), ch = nb_wg * 32 + e % (WB * 32);
            if (ch < cout_real) be.partial[((int64_t)tile * 2 + st) * cout_real + ch] = 0.f;
          }
      }
      return;
    }
  }
  // Slot split (gridDim.z == 3, 3^3 maps of the coarse levels): this workgroup sums only 9 of the 27 offsets into
  // its own fp32 partial image; k_sum_partials adds the three in a fixed order.  A coarse level has too few row
  // tiles to fill the chip and every tile streams ALL weights, so the split buys parallelism, not traffic.
  float *out_f32 = out_f32_arg;
  if (gridDim.z > 1) {
    smask &= 0x1ffu << (9 * blockIdx.z);
    out_f32 += (int64_t)blockIdx.z * zstride;
  }

  f32x16 acc[RB][NCB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;

  // ---- gather indices of every visited slot are parked in LDS once (coalesced), so that the per-offset index
  // fetch is a ds_read (lgkmcnt) and never sits in the VMEM queue in front of the gather ring
  __shared__ __attribute__((aligned(16))) int32_t l_idx[27 * TM];
  // ---- feature-side iterator: flat sequence of (slot, chunk)
  // Order of the reduction: channel GROUPS of gc chunks outermost, then the offsets, then the chunks of the group.
  // gc = nc (one group) is the plain "offset by offset" order.  With wide rows (512 channels = 1 KB) the plain order
  // touches every gathered row in sixteen 64-byte pieces per offset and comes back to it ~14 offsets later: the rows of
  // the co-resident tiles plus 14 MB of weights do not fit the XCD's 4 MB L2 (PMC: L2 hit 51 %, 19x the compulsory HBM
  // bytes); with 128-channel groups a row's 256-byte segment serves all offsets back to back.
  const uint32_t fmask = smask;  // offsets the feature side walks
  uint32_t rem = smask;
  int islot = -1, gbase = 0, gend = min(gc, nc), ichunk = gend;  // forces "advance to first slot" on the first call
  int32_t idx_i[RB];
  auto advance = [&]() __attribute__((always_inline)) -> bool {
    if (++ichunk < gend) return true;
    if (rem == 0) {                       // next channel group: all offsets again
      gbase = gend;
      if (gbase >= nc) return false;
      gend = min(gbase + gc, nc);
      rem = fmask;
    }
    islot = __builtin_ctz(rem);
    rem &= rem - 1;
    ichunk = gbase;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int r = row_of(rb);
      const int64_t p = pos_wg + r;
      idx_i[rb] = v.nbr ? l_idx[islot * TM + r] : (p < v.n_in ? (int32_t)p : -1);
    }
    return true;
  };
  // Gathers and weight fetches are BUFFER loads (scalar descriptor + one 32-bit lane offset): a missing neighbour or
  // a padded channel simply gets an out-of-range offset, for which the hardware returns zeros -- no branches, no
  // 64-bit address arithmetic, no selects (the gather loop was spending ~10 VALU instructions per MFMA on those).
  constexpr unsigned kOOB = 0xfffff000u;   // > any descriptor size accepted by the host wrapper, no wrap with small immediates
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wp), 0, (int)w_bytes, 0x00020000);
  // in_ld = row stride of the gathered tensor in elements (> cin_real when it is a column slice of a wider buffer, e.g.
  // the skip half of a zero-copy ME.cat)
  const unsigned row_bytes = (unsigned)in_ld * (unsigned)sizeof(T);
  const bool ch_tail = (cin_real & 31) != 0;   // kernel-uniform: only then a chunk can run past the row
  auto issue = [&](u32x4 (&F)[RB][LD], uint32_t &act) __attribute__((always_inline)) {
    act = 0;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const bool ok = idx_i[rb] >= 0;
      if (__ballot(idx_i[rb] >= 0)) act |= 1u << rb;
      const unsigned base = ok ? (unsigned)idx_i[rb] * row_bytes + (unsigned)(ichunk * 32 + h * 16) * (unsigned)sizeof(T) : kOOB;
#pragma unroll
      for (int t = 0; t < LD; ++t) {
        unsigned off = base + t * 16;
        if (ch_tail && (ichunk * 32 + h * 16 + t * EPL + EPL > cin_real)) off = kOOB;
        F[rb][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
      }
    }
  };
  // ---- weight-side iterator: (slot, slab) in the same order
  const int nslab = (nc + SC - 1) / SC, gs = (gc + SC - 1) / SC;   // slabs per channel group (gc is a multiple of SC or = nc)
  uint32_t wrem = smask;
  int wslot = -1, wgbase = 0, wgend = min(gs, nslab), wslab = wgend;
  auto wadvance = [&]() __attribute__((always_inline)) -> bool {
    if (++wslab < wgend) return true;
    if (wrem == 0) {
      wgbase = wgend;
      if (wgbase >= nslab) return false;
      wgend = min(wgbase + gs, nslab);
      wrem = smask;
    }
    wslot = __builtin_ctz(wrem);
    wrem &= wrem - 1;
    wslab = wgbase;
    return true;
  };
  // The packed weights are padded to whole slabs (ncp chunks) and whole cout tiles (nbp blocks), zero filled, so a
  // slab is fetched with unconditional, fully coalesced 16-byte buffer loads: per-thread offsets are loop invariant,
  // the slab's base goes in the scalar offset.
  unsigned woff[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int e = min(tid + i * NT, SLAB - 1);
    const int cw = e / WCH, ee = e - cw * WCH;
    woff[i] = (unsigned)(cw * nbp * (WLD * 64) + ee) * 16u;
  }
  auto wissue = [&](u32x4 (&wreg)[WR]) __attribute__((always_inline)) {
    const int kw = v.KS > 1 ? wslot : kw_single;  // 3^3 dgrad mirroring (K-1-k) is folded into the weight packing
    const unsigned sbase = (unsigned)((((int64_t)kw * ncp + wslab * SC) * nbp + nb_wg) * (WLD * 64) * 16);
#pragma unroll
    for (int i = 0; i < WR; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[i], sbase, 0);
  };
  auto wstage = [&](int buf, const u32x4 (&wreg)[WR]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int e = tid + i * NT;
      if (e < SLAB) lds[buf][e] = wreg[i];
    }
  };
  float sumsq = 0.f;   // EPI = 1: |f|^2 of this lane's channel half of row vx
  auto compute = [&](int buf, int cc, const u32x4 (&F)[RB][LD], uint32_t act) __attribute__((always_inline)) {
    if (act == 0) return;
    if constexpr (EPI == 1) {
#pragma unroll
      for (int t = 0; t < LD; ++t) sumsq = sq16<T>(F[0][t], sumsq);
    }
    const u32x4 *wl = &lds[buf][cc * WCH + (wn * NCB) * WLD * 64 + lane];
    if constexpr (Tr<T>::SPLIT) {
      // split fp32: per k-step of 16 channels (two of the lane's four loads) the feature fragment becomes three bf16 operands,
      // reused by all NCB column blocks; the six products go smallest first into the same fp32 accumulator
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        u32x4 wf3[3][NCB];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb) wf3[pc][nb] = wl[(nb * WLD + tt * 3 + pc) * 64];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (act & (1u << rb)) {
            u32x4 fh, fm, fl;
            split8(F[rb][2 * tt], F[rb][2 * tt + 1], fh, fm, fl);
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) mma_bf16(acc[rb][nb], wf3[2][nb], fh);
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) mma_bf16(acc[rb][nb], wf3[0][nb], fl);
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) mma_bf16(acc[rb][nb], wf3[1][nb], fm);
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) mma_bf16(acc[rb][nb], wf3[1][nb], fh);
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) mma_bf16(acc[rb][nb], wf3[0][nb], fm);
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) mma_bf16(acc[rb][nb], wf3[0][nb], fh);
          }
        }
      }
      return;
    }
    // all weight fragments of the chunk are requested up front: the LDS latency of step t+1 hides under the
    // MFMAs of step t (the compiler otherwise waits lgkmcnt(0) in front of every MFMA triple)
    u32x4 wf[LD][NCB];
#pragma unroll
    for (int t = 0; t < LD; ++t)
#pragma unroll
      for (int nb = 0; nb < NCB; ++nb) wf[t][nb] = wl[(nb * LD + t) * 64];
#pragma unroll
    for (int t = 0; t < LD; ++t) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (act & (1u << rb)) {
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb) mma16<T>(acc[rb][nb], wf[t][nb], F[rb][t]);
        }
      }
    }
  };

  u32x4 F[D][RB][LD], wreg[WR];
  uint32_t act[D];
  int buf = 0, cc = 0;       // LDS buffer holding the current slab, chunk index inside it
  int ncs = 0;               // chunks in the current slab
  bool wnext = false;        // is a following slab prefetched in wreg?
  int issued = 0, computed = 0;
  int pslab = -1;               // slab index of the prefetched weight slab
  // Prologue: the first weight slab and ALL index loads of the tile are in flight together, one barrier publishes
  // both (a slot-by-slot index copy loop was a chain of ~16 dependent global-load latencies at the head of every
  // workgroup -- a quarter of its lifetime -- and the first weight fetch only started behind it).
  const bool whave = wadvance();
  if (whave) wissue(wreg);
  // kernel-map rows of the tile: SIXTEEN-byte loads, four consecutive positions of one offset per lane (the table is
  // offset-major and position tiles are 16-byte aligned): 27 x TM / 4 / 64 wave-instructions per tile instead of one 4-byte
  // load per (offset, wave) -- 27 instead of 108 per 256-position tile of the ~900 vector-memory instructions the kernel is
  // bound by.  Offsets the tile does not visit get an out-of-range address (zeros, never read).
  const uint64_t idx_bytes = (uint64_t)v.KS * (uint64_t)v.n_pad * 4ull;
  if (v.nbr && idx_bytes < 0xfffff000ull) {     // kernel-uniform
    constexpr int Q = TM / 4, NU = (27 * Q + NT - 1) / NT;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(v.nbr), 0, (int)idx_bytes, 0x00020000);
    u32x4 t4[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = tid + i * NT, sl = e / Q, q = e - sl * Q;
      const bool ok = sl < v.KS && ((smask >> sl) & 1u);
      const unsigned off = ok ? (unsigned)(((int64_t)sl * v.n_pad + pos_wg + 4 * q) * 4) : 0xfffff000u;
      t4[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = tid + i * NT, sl = e / Q, q = e - sl * Q;
      if (sl < 27) *reinterpret_cast<u32x4 *>(&l_idx[sl * TM + 4 * q]) = t4[i];
    }
  } else if (v.nbr) {
    constexpr int IT = (TM + NT - 1) / NT;
    int32_t tmp[27][IT];
#pragma unroll
    for (int sl = 0; sl < 27; ++sl) {
      if ((smask >> sl) & 1u) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int r = tid + it * NT;
          tmp[sl][it] = r < TM ? v.nbr[(int64_t)sl * v.n_pad + pos_wg + r] : -1;
        }
      }
    }
#pragma unroll
    for (int sl = 0; sl < 27; ++sl) {
      if ((smask >> sl) & 1u) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int r = tid + it * NT;
          if (r < TM) l_idx[sl * TM + r] = tmp[sl][it];
        }
      }
    }
  }
  if (whave) {
    wstage(0, wreg);
    ncs = min(SC, nc - wslab * SC);
    wnext = wadvance();
    if (wnext) { pslab = wslab; wissue(wreg); }
  }
  __syncthreads();   // indices and the first weight slab are visible
  const int total = __builtin_popcount(fmask) * nc;  // chunks of this wave
#pragma unroll
  for (int d = 0; d < D - 1; ++d) {
    act[d] = 0;
    if (issued < total) { advance(); issue(F[d], act[d]); ++issued; }
  }
  act[D - 1] = 0;
  // at the end of a slab publish the prefetched next slab and cross one barrier
  auto slab_end = [&]() __attribute__((always_inline)) {
    if (wnext) wstage(buf ^ 1, wreg);
    __syncthreads();
    buf ^= 1;
    cc = 0;
    if (wnext) {
      ncs = min(SC, nc - pslab * SC);
      wnext = wadvance();
      if (wnext) { pslab = wslab; wissue(wreg); }
    }
  };
  // steady state: every sub-step issues one chunk and computes one chunk UNCONDITIONALLY, so the compiler can
  // count outstanding loads (s_waitcnt vmcnt(N), N > 0) instead of draining the queue before every MFMA group
  while (total - issued >= D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      advance();
      issue(F[(d + D - 1) % D], act[(d + D - 1) % D]);
      compute(buf, cc, F[d], act[d]);
      if (++cc == ncs) slab_end();
    }
    issued += D;
    computed += D;
  }
  // tail (< 2D chunks)
  while (computed < total) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (computed < total) {
        if (issued < total) { advance(); issue(F[(d + D - 1) % D], act[(d + D - 1) % D]); ++issued; }
        compute(buf, cc, F[d], act[d]);
        ++computed;
        if (++cc == ncs) slab_end();
      }
    }
  }

  if constexpr (EPI == 1) {
    // ---- CLIP-loss epilogue.  The weight LDS is idle now: every wave parks one 32 x 32 block of its similarity tile
    // there at a time (row stride 36 floats) so that a lane can pick the entries of ITS row's label / negatives with
    // an indexed ds_read (registers cannot be indexed per lane); the arg-max runs on the registers.
    __syncthreads();
    float *tile = reinterpret_cast<float *>(&lds[0][0]) + wave * (32 * 36);
    sumsq += __shfl_xor(sumsq, 32);
    const float inv = 1.f / fmaxf(sqrtf(sumsq), 1e-12f);
    const int64_t p = pos_w + vx;
    const bool live = p < v.n_out;
    // half h of row vx handles targets j = h, h + 2, ...  (j = 0: the positive, j >= 1: negative j - 1)
    int tcol[4];
    int64_t lab = ce.ignore;
    if (live) lab = ce.labels[p];
    const bool valid = live && lab != ce.ignore && lab >= 0 && lab < ce.n_anchor;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 2 * jj + h;
      int t = -1;
      if (valid && j <= ce.k_neg) {
        const int64_t tt = j == 0 ? lab : ce.neg[p * ce.k_neg + (j - 1)];
        t = (tt >= 0 && tt < ce.n_anchor) ? (int)tt : -1;
      }
      tcol[jj] = t;
    }
    float best = -3.0e38f, spos = 0.f, sneg = 0.f;
    int bi = 0;
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) {
      if (nb_w + nb >= nb_total) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = (nb_w + nb) * 32 + 8 * q + 4 * h;
        const float s0 = acc[0][nb][4 * q + 0] * inv, s1 = acc[0][nb][4 * q + 1] * inv, s2 = acc[0][nb][4 * q + 2] * inv,
                    s3 = acc[0][nb][4 * q + 3] * inv;
        *reinterpret_cast<float4 *>(tile + vx * 36 + 8 * q + 4 * h) = make_float4(s0, s1, s2, s3);
        if (c0 + 0 < ce.n_anchor && s0 > best) { best = s0; bi = c0 + 0; }
        if (c0 + 1 < ce.n_anchor && s1 > best) { best = s1; bi = c0 + 1; }
        if (c0 + 2 < ce.n_anchor && s2 > best) { best = s2; bi = c0 + 2; }
        if (c0 + 3 < ce.n_anchor && s3 > best) { best = s3; bi = c0 + 3; }
        if (out_f32 && live && c0 < cout_real)
          *reinterpret_cast<float4 *>(out_f32 + p * cout_real + c0) = make_float4(s0, s1, s2, s3);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int t = tcol[jj];
        if (t >= 0 && (t >> 5) == nb_w + nb) {
          const float sv = tile[vx * 36 + (t & 31)];
          if (jj == 0 && h == 0) spos = sv; else sneg += sv;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    {  // the two halves of a row: first maximum (lowest index on ties), sums
      const float ob = __shfl_xor(best, 32);
      const int oi = __shfl_xor(bi, 32);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      sneg += __shfl_xor(sneg, 32);
      spos += __shfl_xor(spos, 32);
    }
    if (live && h == 0) {
      ce.d_pos[p] = valid ? 1.f - spos : 0.f;
      ce.d_neg[p] = valid ? 1.f - sneg / (float)ce.k_neg : 0.f;
      if (ce.pred) ce.pred[p] = bi;
      if (ce.inv_norm) ce.inv_norm[p] = inv;
    }
    return;
  }
  // ---- epilogue: lane (voxel vx, half h) owns channels nb*32 + 8q + 4h + {0..3}
  // bf16 rows on the 8-channel grid: SIXTEEN-byte stores.  The accumulator layout gives a lane four 4-channel pieces (8 bytes of
  // bf16 each) 16 bytes apart; the two half-wave lanes of a voxel hold interleaved pieces.  v_permlane32_swap trades piece
  // q = 2p+1 of the h = 0 lane for piece q = 2p of the h = 1 lane, so that lane h owns the 8 contiguous channels
  // nb*32 + 16p + 8h .. + 7: half as many store instructions (the kernel sits on the CU's vector-memory instruction rate, and
  // the classifier's 200-channel rows are mostly stores).  Same values, same rounding -- only who writes them changes.
  if constexpr (EPL == 8 && EPI == 0) {
    if (!out_f32 && (cout_real & 7) == 0) {      // kernel-uniform
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        int64_t p = pos_wg + row_of(rb);
        int32_t orow = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
        T *dst = out + (int64_t)(orow < 0 ? 0 : orow) * cout_real;
#pragma unroll
        for (int nb = 0; nb < NCB; ++nb) {
          if (nb_w + nb >= nb_total) continue;
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            uint32_t pk[2][2];                   // [piece q = 2pq, 2pq+1][dword]
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int q = 2 * pq + e, c0 = (nb_w + nb) * 32 + 8 * q + 4 * h;
              float o0 = acc[rb][nb][4 * q + 0], o1 = acc[rb][nb][4 * q + 1], o2 = acc[rb][nb][4 * q + 2], o3 = acc[rb][nb][4 * q + 3];
              if (bias && c0 < cout_real) { o0 += bias[c0]; o1 += bias[c0 + 1]; o2 += bias[c0 + 2]; o3 += bias[c0 + 3]; }
              pk[e][0] = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);
              pk[e][1] = (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16);
            }
            // after the swaps: lane h holds [pk0 | pk1] = channels nb*32 + 16 pq + 8h + {0..3 | 4..7}
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const int c8 = (nb_w + nb) * 32 + 16 * pq + 8 * h;
            if (orow >= 0 && c8 < cout_real) {
              u32x4 val = {s0[0], s1[0], s0[1], s1[1]};
              if (be.accum) {     // kernel-uniform: rounded exactly like "store the result, then add the two tensors"
                const u32x4 prev = *reinterpret_cast<const u32x4 *>(dst + c8);
                const uint32_t a[4] = {val.x, val.y, val.z, val.w}, b[4] = {prev.x, prev.y, prev.z, prev.w};
                uint32_t r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
                  const float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
                  r[i] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
                }
                val = u32x4{r[0], r[1], r[2], r[3]};
              }
              *reinterpret_cast<u32x4 *>(dst + c8) = val;
            }
          }
        }
      }
      goto stats;
    }
  }
  // ---- epilogue: lane (voxel vx, half h) owns channels nb*32 + 8q + 4h + {0..3}
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int64_t p = pos_wg + row_of(rb);
    int32_t orow = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
    if (orow < 0) continue;
    T *dst = out + (int64_t)orow * cout_real;
    const float rs = (out_f32 && row_scale) ? row_scale[orow] : 1.f;
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) {
      if (nb_w + nb >= nb_total) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int c0 = (nb_w + nb) * 32 + 8 * q + 4 * h;
        if (c0 >= cout_real) continue;
        float o0 = acc[rb][nb][4 * q + 0], o1 = acc[rb][nb][4 * q + 1], o2 = acc[rb][nb][4 * q + 2],
              o3 = acc[rb][nb][4 * q + 3];
        if (bias) { o0 += bias[c0]; o1 += bias[c0 + 1]; o2 += bias[c0 + 2]; o3 += bias[c0 + 3]; }
        if constexpr (EPI == 0) {
          if (be.accum) {   // kernel-uniform.  The sum is rounded exactly like "store the result, then add the two tensors"
            float prev[4];
            load4(dst + c0, prev);
            o0 = stored_value<T>(o0) + prev[0]; o1 = stored_value<T>(o1) + prev[1];
            o2 = stored_value<T>(o2) + prev[2]; o3 = stored_value<T>(o3) + prev[3];
          }
        }
        if (out_f32) {  // CLIP similarity: fp32 output, per-row scale (1/|f|)
          *reinterpret_cast<float4 *>(out_f32 + (int64_t)orow * cout_real + c0) = make_float4(o0 * rs, o1 * rs, o2 * rs, o3 * rs);
        } else if constexpr (EPL == 4) {
          *reinterpret_cast<float4 *>(dst + c0) = make_float4(o0, o1, o2, o3);
        } else {
          uint2 pk;
          pk.x = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);
          pk.y = (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16);
          *reinterpret_cast<uint2 *>(dst + c0) = pk;
        }
      }
    }
  }
stats:
  if constexpr (EPI == 0) {
    if (be.partial != nullptr) {   // kernel-uniform
      // ---- BatchNorm statistics of this workgroup's rows.  Lane (vx, h) holds, per row block, the 16 NCB channels
      // nb*32 + 8q + 4h + i of voxel vx: sum the row blocks in registers, then a HALVING butterfly over the 32 lanes of a
      // half wave (step k: a lane keeps one half of its values and adds the partner's copy of that half) -- V values cost
      // ~V shuffles instead of 5 V -- leaves every lane with NCB finished column sums; the WM waves are folded through
      // the (now idle) weight LDS in wave order, i.e. deterministically.
      constexpr int V = 32 * NCB;              // [stat][nb][r]: 16 NCB sums + 16 NCB sums of squares
      float vst[V];
#pragma unroll
      for (int j = 0; j < V; ++j) vst[j] = 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int64_t p = pos_wg + row_of(rb);
        const int32_t orow = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
        const bool live = orow >= 0;
#pragma unroll
        for (int nb = 0; nb < NCB; ++nb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = (nb_w + nb) * 32 + 8 * q + 4 * h;
            float pv[4] = {0.f, 0.f, 0.f, 0.f};
            if (be.pivot && c0 < cout_real) { const float4 t4 = *reinterpret_cast<const float4 *>(be.pivot + c0); pv[0] = t4.x; pv[1] = t4.y; pv[2] = t4.z; pv[3] = t4.w; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float o = acc[rb][nb][4 * q + i];
              if (bias && c0 + i < cout_real) o += bias[c0 + i];
              const float d = (live && c0 < cout_real) ? stored_value<T>(o) - pv[i] : 0.f;
              vst[nb * 16 + 4 * q + i] += d;
              vst[16 * NCB + nb * 16 + 4 * q + i] += d * d;
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int n = V >> k;                  // values held before this step
        const bool up = (vx >> k) & 1;         // this lane keeps the upper half
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
          const float keep = up ? vst[n / 2 + i] : vst[i];
          const float send = up ? vst[i] : vst[n / 2 + i];
          vst[i] = keep + __shfl_xor(send, 1 << k);
        }
      }
      // lane (vx, h) now holds original indices j = sum_k bit_k(vx) * (V >> (k+1)) + t, t < NCB
      __syncthreads();                         // every wave is done with the weight LDS
      float *stage = reinterpret_cast<float *>(&lds[0][0]);     // [WM][2][WB*32]
      constexpr int CT = WB * 32;
      int jbase = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) jbase += ((vx >> k) & 1) * (V >> (k + 1));
#pragma unroll
      for (int t = 0; t < NCB; ++t) {
        const int j = jbase + t, st = j / (16 * NCB), jj = j % (16 * NCB), nb = jj / 16, r = jj % 16;
        const int cl = (wn * NCB + nb) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);       // channel inside the workgroup's tile
        stage[(wm * 2 + st) * CT + cl] = vst[t];
      }
      __syncthreads();
      for (int e = tid; e < 2 * CT; e += NT) {
        const int st = e / CT, cl = e % CT, ch = nb_wg * 32 + cl;
        if (ch < cout_real) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) sum += stage[(w * 2 + st) * CT + cl];
          be.partial[((int64_t)tile * 2 + st) * cout_real + ch] = sum;
        }
      }
    }
  }
}

// rows that no position of the view writes must still be defined: the forward output of a strided
// map always covers every row, but a grouped view's padding never does -- nothing to do there.

// ------------------------------------------------------------------------------------ host side

// Tile choice: 256 positions x up to 128 channels when the map fills the chip, otherwise 128-position x 64-channel
// tiles, one 32 x 64 block per wave (coarse levels: few rows, many channels).
// SC = chunks per weight slab (sized to ~6-8 staging registers per thread), D = depth of the gather ring.
struct GatherCfg { int id, sc, wb, tm; };   // tm = positions per workgroup tile
template <typename T>
GatherCfg gather_cfg(const View &v, int nb_total) {
  constexpr bool kF32 = (sizeof(T) == 4);
  const bool big = v.n_pad >= 256 * 256;
  if (big) {
    // 5..7 blocks (e.g. the 200 classes / 200 CLIP anchors = 7 blocks): one 128-position tile spans ALL output
    // channels, so the [N, C] feature matrix is streamed exactly once (dense GEMM with a small N)
    if (nb_total >= 5 && nb_total <= 7 && v.nbr == nullptr && tune(T_HEAD_TILE) == 0) return {6, kF32 ? 1 : 2, 7, 128};
    if (nb_total == 1) return {0, kF32 ? 2 : 4, 1, 256};
    if (nb_total == 2) return {1, kF32 ? 2 : 4, 2, 256};
    if (nb_total == 3 || (nb_total % 3 == 0 && nb_total % 4 != 0)) return {2, kF32 ? 2 : 4, 3, 256};
    // wide outputs (>= 256 channels, e.g. the 512-d CLIP representation model): id 17 = the 2-D blocked LDS-DMA kernel of
    // lgs_conv_wide.hip (round 3; L0 512 -> 512: 9.4 ms); id 16 = eight waves of 32 positions x 256 channels on this kernel
    // (round 2: 12.0 ms; still what 1x1 layers below 512 output channels and CONV_WIDE=0 take)
    if (!kF32 && tune(T_CONV_WIDE) != 0 && (nb_total % 8 == 0 || nb_total >= 16) && !(v.KS == 1 && v.nbr == nullptr && nb_total < 16))
      return {17, 2, 8, 256};
    if (!kF32 && nb_total % 8 == 0) return {16, 2, 8, 256};
    if (!kF32) return {7, 2, 4, 128};   // bf16: 128-position tiles, 4 column blocks per wave at 3 waves/SIMD
    return {3, kF32 ? 1 : 2, 4, 256};
  }
  if (nb_total == 1) return {4, kF32 ? 2 : 4, 1, 64};
  const int small_override = (int)tune(T_SMALL_CFG);  // tuning knob
  if (!kF32 && small_override == 5) return {5, 4, 2, 64};
  if (!kF32 && small_override == 9) return {9, 4, 4, 64};
  if (!kF32 && small_override == 10) return {10, 4, 2, 64};
  if (!kF32 && small_override == 11) return {11, 4, 4, 64};
  if (!kF32 && small_override == 3 && nb_total % 4 == 0) return {3, 2, 4, 256};     // 256 positions x 128 channels (64 x 128 per wave)
  if (!kF32 && small_override == 7 && nb_total % 4 == 0) return {7, 2, 4, 128};     // 128 positions x 128 channels (32 x 128 per wave)
  if (!kF32 && small_override == 12) return {12, 8, 2, 128};
  if (!kF32 && small_override == 13) return {13, 8, 4, 128};
  // 3^3 maps of 16 k+ positions with 256 output channels (level 3 of the 8-scene batch): 128 positions x 128 channels per workgroup
  // gathers every row half as often as the 64-channel tile (stand-alone 256 -> 256 at 19.6 k rows: 0.127 vs 0.147 ms; the other
  // coarse shapes -- level 3 128 -> 128, level 4 256 -> 256 at 5 k rows -- are faster on the small tile, r04_experiments.txt)
  if (!kF32 && nb_total == 8 && v.KS > 1 && v.n_pad >= 16384) return {7, 2, 4, 128};
  if (!kF32) return {8, 4, 2, 128};   // measured best on the other L3/L4 shapes (tools/microbench.py coarse): 128 positions x 64 channels
  return {5, 2, 2, 64};
}

// out = p0 + p1 + p2 (+ bias), fixed order; 4 elements per thread
template <typename T>
__global__ void k_sum_partials(const float *__restrict__ part, int64_t n4, int64_t zstride, const float *__restrict__ bias,
                               int cout, T *__restrict__ out, int accum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4 *>(part)[i];
  const float4 b = reinterpret_cast<const float4 *>(part + zstride)[i];
  const float4 c = reinterpret_cast<const float4 *>(part + 2 * zstride)[i];
  float o0 = a.x + b.x + c.x, o1 = a.y + b.y + c.y, o2 = a.z + b.z + c.z, o3 = a.w + b.w + c.w;
  if (bias) { const int c0 = (int)((i * 4) % cout); o0 += bias[c0]; o1 += bias[c0 + 1]; o2 += bias[c0 + 2]; o3 += bias[c0 + 3]; }
  if (accum) {
    float prev[4];
    load4(out + 4 * i, prev);
    o0 = stored_value<T>(o0) + prev[0]; o1 = stored_value<T>(o1) + prev[1];
    o2 = stored_value<T>(o2) + prev[2]; o3 = stored_value<T>(o3) + prev[3];
  }
  if constexpr (sizeof(T) == 4) {
    reinterpret_cast<float4 *>(out)[i] = make_float4(o0, o1, o2, o3);
  } else {
    uint2 pk;
    pk.x = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);
    pk.y = (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16);
    reinterpret_cast<uint2 *>(out)[i] = pk;
  }
}

constexpr int64_t kSplitMaxBytes = 16ll << 20;   // fp32 partial images of the slot split (larger ones cost more than the split gains)
inline int64_t split_partial_bytes(int K, int64_t n_out, int o_real) {
  if (K != 27 || n_out <= 0) return 0;
  const int64_t b = 3 * n_out * (int64_t)o_real * 4;
  return b <= kSplitMaxBytes ? align256(b) : 0;
}

// epilogue options of one launch: a slot-split launch writes fp32 partial images (statistics / accumulation happen in
// k_sum_partials or not at all)
inline BnEpi bn_epi(const BnEpi *bn, bool did_split) {
  BnEpi e = (bn && !did_split) ? *bn : BnEpi();
  return e;
}

template <typename T>
int launch_gather(const View &v, const GatherCfg &cfg, const T *in, int cin_real, int nc, const uint4 *wp,
                  int nb_total, int ncp, int nbp, int K, T *out, int cout_real, const float *bias, hipStream_t s,
                  float *out_f32 = nullptr, const float *row_scale = nullptr, float *zpartial = nullptr,
                  const BnEpi *bn = nullptr, int *bn_rows = nullptr, int in_ld = 0) {
  if (v.n_pad == 0) return 0;
  constexpr int LDc = Tr<T>::WLD;
  const uint64_t in_bytes64 = (uint64_t)v.n_in * (uint64_t)(in_ld > 0 ? in_ld : cin_real) * sizeof(T);
  const uint64_t w_bytes64 = (uint64_t)K * ncp * nbp * LDc * 64 * 16;
  LGS_REQUIRE(in_bytes64 < 0xfffff000ull && w_bytes64 < 0xfffff000ull,
              "sparse conv: a feature or weight tensor of 4 GiB or more is beyond the 32-bit buffer-descriptor path");
  const unsigned in_bytes = (unsigned)in_bytes64, w_bytes = (unsigned)w_bytes64;
  // bf16 storage only: the fp32 path is the parity mode and keeps ONE accumulator per output (a different summation
  // order moves results by ~1e-7, which BatchNorm over a handful of coarse rows with near-zero variance amplifies into
  // ReLU gate flips against the oracle -- measured on the 14A fixture)
  const bool no_split = tune(T_CONV_SPLIT) == 0;   // debugging knob
  const bool can_split = !no_split && sizeof(T) == 2 && zpartial && !out_f32 && v.KS > 1 && K == 27 && split_partial_bytes(K, v.n_out, cout_real) > 0;
  const int64_t zstride = v.n_out * (int64_t)cout_real;
  bool did_split = false;
  constexpr bool kF32 = (sizeof(T) == 4);
  // channel groups of the reduction (see the kernel): rows wider than 8 chunks (256 channels) are walked in 4-chunk groups
  const int gc = nc > 8 ? 4 : nc;
#define LGS_LAUNCH(RB, NCB, WM, WN, SC, D)                                                                        \
  do {                                                                                                            \
    dim3 grid((unsigned)(v.n_pad / (WM * RB * 32)), (unsigned)((nb_total + WN * NCB - 1) / (WN * NCB)));         \
    did_split = can_split && (int64_t)grid.x * grid.y < 600;   /* the chip holds >= 512 of these workgroups */    \
    if (did_split) grid.z = 3;                                                                                    \
    LGS_KLAUNCH((k_conv_gather<T, RB, NCB, WM, WN, SC, D>), grid, dim3(WM *WN * 64), 0, s, v, in, cin_real, nc,    \
                       reinterpret_cast<const u32x4 *>(wp), nb_total, ncp, nbp, out, cout_real, did_split ? nullptr : bias, \
                       did_split ? zpartial : out_f32, row_scale, in_bytes, w_bytes, zstride, ClipEpi(),           \
                       bn_epi(bn, did_split), gc, in_ld > 0 ? in_ld : cin_real);                                  \
    if (bn_rows) *bn_rows = did_split ? 0 : (int)grid.x;                                                          \
  } while (0)
  if (cfg.id == 17) {
    LGS_REQUIRE(sizeof(T) == 2 && !out_f32 && !row_scale && !(bn && bn->partial), "wide conv: bf16 feature output only (internal error)");
    LGS_REQUIRE(!(bn && bn->accum), "wide conv: no accumulating epilogue (lgs_conv_dgrad_can_accumulate says so)");
    if (bn_rows) *bn_rows = 0;
    return launch_conv_wide(v, in, cin_real, in_ld > 0 ? in_ld : cin_real, wp, nb_total, ncp, nbp, K, out, cout_real, bias,
                            gc >= nc ? 0 : (gc + 1) / 2, s);
  }
  switch (cfg.id) {
    case 0: LGS_LAUNCH(2, 1, 4, 1, (kF32 ? 2 : 4), (kF32 ? 3 : 4)); break;
    case 1: LGS_LAUNCH(2, 2, 4, 1, (kF32 ? 2 : 4), (kF32 ? 3 : 4)); break;
    case 2: LGS_LAUNCH(2, 3, 4, 1, (kF32 ? 2 : 4), (kF32 ? 3 : 4)); break;
    case 3: LGS_LAUNCH(2, 4, 4, 1, (kF32 ? 1 : 2), (kF32 ? 2 : 3)); break;
    case 4: LGS_LAUNCH(1, 1, 2, 1, (kF32 ? 2 : 4), (kF32 ? 4 : 8)); break;
    case 6: LGS_LAUNCH(1, 7, 4, 1, (kF32 ? 1 : 2), (kF32 ? 3 : 4)); break;
    case 7: if constexpr (!kF32) LGS_LAUNCH(1, 4, 4, 1, 2, 4); break;      // ids 7 .. 13, 16: bf16 only (gather_cfg)
    case 16: if constexpr (!kF32) LGS_LAUNCH(1, 8, 8, 1, 2, 4); break;
    case 8: if constexpr (!kF32) LGS_LAUNCH(1, 2, 4, 1, 4, 6); break;
    case 9: if constexpr (!kF32) LGS_LAUNCH(1, 2, 2, 2, 4, 6); break;
    case 10: if constexpr (!kF32) LGS_LAUNCH(1, 2, 2, 1, 4, 6); break;
    case 11: if constexpr (!kF32) LGS_LAUNCH(1, 4, 2, 1, 4, 6); break;
    case 12: if constexpr (!kF32) LGS_LAUNCH(1, 2, 4, 1, 8, 6); break;     // as 8 with 8-chunk (256-channel) weight slabs: half the slab barriers
    case 13: if constexpr (!kF32) LGS_LAUNCH(1, 4, 4, 1, 8, 4); break;     // 128 positions x 128 channels, 8-chunk slabs
    default: LGS_LAUNCH(1, 1, 2, 2, (kF32 ? 2 : 4), (kF32 ? 4 : 8)); break;
  }
#undef LGS_LAUNCH
  if (did_split) {
    const int64_t n4 = zstride / 4;
    if (n4 > 0) LGS_KLAUNCH((k_sum_partials<T>), (unsigned)((n4 + 255) / 256), 256, 0, s, zpartial, n4, zstride, bias, cout_real, out,
                                   (bn && bn->accum) ? 1 : 0);
  }
  LGS_HIP(hipGetLastError());
  return 0;
}

// rows of BatchNorm statistics the forward launch of this shape writes (= its position tiles), 0 if the launch cannot
// produce them (slot-split launches sum partial images afterwards; odd output widths go through a scratch image)
template <typename T>
int bn_partial_rows_t(const View &v, int K, int o_real) {
  if (v.n_pad == 0 || o_real % 4 != 0) return 0;
  const int nb_total = pad32(o_real) / 32;
  const GatherCfg cfg = gather_cfg<T>(v, nb_total);
  if (cfg.id == 17) return 0;        // the wide kernel has no statistics epilogue
  const int64_t gx = v.n_pad / cfg.tm, gy = (nb_total + cfg.wb - 1) / cfg.wb;
  const bool no_split = tune(T_CONV_SPLIT) == 0;
  const bool split = sizeof(T) == 2 && !no_split && v.KS > 1 && K == 27 &&
                     split_partial_bytes(K, v.n_out, o_real) > 0 && gx * gy < 600;
  return split ? 0 : (int)gx;
}

// T = storage type of the tensors, TK = the kernel instance that multiplies them (TK = f32s_t: fp32 tensors, split-bf16 products)
template <typename T, typename TK = T>
int conv_gather_op(const View &v, const void *in_v, int g_real, const float *weight, int K, int cin_w, int cout_w,
                   int transposed_w, int o_real, const float *bias, void *out_v, void *workspace, hipStream_t s,
                   int w_o_real = -1, const BnEpi *bn = nullptr, void *packed_ext = nullptr, int pack_mode = 0, int in_ld = 0) {
  if (w_o_real < 0) w_o_real = o_real;
  if (v.n_pad == 0) return 0;      // the maps of an empty batch: no row to write (and a 256-byte workspace: nothing is packed)
  if constexpr (std::is_same<TK, bf16_t>::value) {
    // 1x1 layers of the big maps: a streaming GEMM with persistent workgroups (lgs_pointwise.hip), no packed image, no workspace
    const int64_t ld = in_ld > 0 ? in_ld : g_real;
    if (!bn && w_o_real == o_real && pointwise_supported(v, K, g_real, o_real, ld))
      return launch_pointwise(v, in_v, ld, g_real, weight, cin_w, cout_w, transposed_w, o_real, bias, out_v, s);
  }
  LGS_REQUIRE(in_ld == 0 || in_ld == g_real || (g_real % Tr<T>::EPL == 0 && o_real % 4 == 0 && in_ld > g_real && (in_ld * (int)sizeof(T)) % 16 == 0),
              "sparse conv: a strided input needs 16-byte aligned rows and channel counts on the 16-byte grid");
  constexpr int EPL = Tr<T>::EPL, LD = Tr<TK>::WLD;
  const int g_pad = pad32(g_real), nc = g_pad / 32, nb_total = pad32(o_real) / 32;
  char *ws = reinterpret_cast<char *>(workspace);
  uint4 *wp = reinterpret_cast<uint4 *>(ws);
  const GatherCfg cfg = gather_cfg<TK>(v, nb_total);
  const int ncp = (nc + cfg.sc - 1) / cfg.sc * cfg.sc, nbp = (nb_total + cfg.wb - 1) / cfg.wb * cfg.wb;
  int64_t wbytes = align256((int64_t)K * (nc + 3) * (nb_total + 3) * LD * 64 * 16);
  if (o_real % 4 != 0) {
    // rows are written in 4-channel groups: route odd widths (e.g. the 3-channel input gradient of a
    // test) through a 4-aligned scratch image placed after the packed weights and the padded input
    const int o4 = (o_real + 3) / 4 * 4;
    int64_t off = wbytes + ((g_real % EPL != 0) ? align256(v.n_in * (int64_t)g_pad * (int64_t)sizeof(T)) : 0);
    T *tmp = reinterpret_cast<T *>(ws + off);
    if (v.n_out > 0) LGS_HIP(hipMemsetAsync(tmp, 0, (size_t)v.n_out * o4 * sizeof(T), s));
    const float *bias4 = nullptr;
    if (bias) {   // e.g. the 3-channel offset head of the instance-segmentation model: bias padded to the scratch width
      float *bp = reinterpret_cast<float *>(ws + off + align256(v.n_out * (int64_t)o4 * (int64_t)sizeof(T)));
      LGS_HIP(hipMemsetAsync(bp, 0, sizeof(float) * o4, s));
      LGS_HIP(hipMemcpyAsync(bp, bias, sizeof(float) * o_real, hipMemcpyDeviceToDevice, s));
      bias4 = bp;
    }
    int rc = conv_gather_op<T, TK>(v, in_v, g_real, weight, K, cin_w, cout_w, transposed_w, o4, bias4, tmp, workspace, s, o_real);
    if (rc) return rc;
    int64_t tot = v.n_out * (int64_t)o_real;
    if (tot > 0)
      LGS_KLAUNCH((k_unpad_rows<T>), (unsigned)((tot + 255) / 256), 256, 0, s, tmp, v.n_out, o_real, o4,
                         reinterpret_cast<T *>(out_v));
    LGS_HIP(hipGetLastError());
    return 0;
  }
  if constexpr (std::is_same<T, float>::value) {
    // fp32 1x1 layers of the big maps: streaming GEMM on the exact-fp32 MFMA (k_pointwise_f32), whichever instance multiplies the 3^3 layers
    const int64_t ld = in_ld > 0 ? in_ld : g_real;
    if (!bn && w_o_real == o_real && pointwise_f32_supported(v, K, g_real, o_real, ld))
      return launch_pointwise_f32(v, in_v, ld, g_real, weight, cin_w, cout_w, transposed_w, o_real, bias, out_v, s);
  }
  const T *in = reinterpret_cast<const T *>(in_v);
  int g_stride = g_real;
  if (g_real % EPL != 0) {  // e.g. the 3-channel colour input of conv0p1s1
    // rows are padded to ONE 16-byte piece multiple (3 -> 8 bf16 / 4 fp32 channels), not to the 32-channel chunk: the
    // pieces beyond the row are masked by the gather's channel-tail test, so neighbours cost 16 instead of 64 bytes
    T *padded = reinterpret_cast<T *>(ws + wbytes);
    const int g_al = (g_real + EPL - 1) / EPL * EPL;
    int64_t tot = v.n_in * g_al;
    if (tot > 0) LGS_KLAUNCH((k_pad_rows<T>), (unsigned)((tot + 255) / 256), 256, 0, s, in, v.n_in, g_real, g_al, padded);
    in = padded;
    g_stride = g_al;
  }
  int64_t total = (int64_t)K * ncp * nbp * LD * 64;
  // packed_ext: a caller-owned image of exactly this layout (lgs_conv_pack_desc); pack_mode 2 = it is up to date
  if (packed_ext && o_real % 4 == 0 && g_real % EPL == 0) wp = reinterpret_cast<uint4 *>(packed_ext);
  else pack_mode = 0;
  if (pack_mode != 2)
    LGS_KLAUNCH((k_pack_weights<TK>), (unsigned)((total + 255) / 256), 256, 0, s, weight, K, cin_w, cout_w, transposed_w,
                       v.mirror, g_real, w_o_real, ncp, nbp, wp);
  LGS_HIP(hipGetLastError());
  // fp32 partial images of the slot split live behind the packed weights and the padded input
  float *zpartial = nullptr;
  if (w_o_real == o_real && split_partial_bytes(K, v.n_out, o_real) > 0)
    zpartial = reinterpret_cast<float *>(ws + wbytes + ((g_real % EPL != 0) ? align256(v.n_in * (int64_t)g_pad * (int64_t)sizeof(T)) : 0));
  int rows = 0;
  int rc = launch_gather<TK>(v, cfg, reinterpret_cast<const TK *>(in), g_stride, nc, wp, nb_total, ncp, nbp, K, reinterpret_cast<TK *>(out_v), o_real, bias, s,
                            nullptr, nullptr, zpartial, bn, &rows, (in_ld > g_real && g_stride == g_real) ? in_ld : 0);
  if (rc) return rc;
  LGS_REQUIRE(!(bn && bn->partial) || rows == bn_partial_rows_t<TK>(v, K, o_real),
              "conv forward: BatchNorm statistics rows differ from lgs_conv_bn_partial_rows (internal error)");
  return 0;
}

// ------------------------------------------------------------------------------------ CLIP contraction
// S = normalize(F) . normalize(T)^T is the 1x1 "convolution" of the voxel features with the
// normalised text anchors as the weight matrix, scaled per row by 1/|f|: it reuses the MFMA gather
// kernel with the identity view.  (ContrastiveLanguageLoss.py:73-95, lib/losses/utils.py:80-103)
template <typename T>
__global__ void k_row_invnorm(const T *__restrict__ f, int64_t n, int c, float *__restrict__ inv) {
  // one wavefront per row
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  float s = 0.f;
  for (int ch = lane; ch < c; ch += 64) { float x = ld_elem(f + row * c + ch); s += x * x; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) inv[row] = 1.f / fmaxf(sqrtf(s), 1e-12f);
}
__global__ void k_normalize_anchors(const float *__restrict__ a, int na, int c, float *__restrict__ out) {
  int row = blockIdx.x;
  int lane = threadIdx.x;
  if (row >= na) return;
  float s = 0.f;
  for (int ch = lane; ch < c; ch += 64) { float x = a[(int64_t)row * c + ch]; s += x * x; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int ch = lane; ch < c; ch += 64) out[(int64_t)row * c + ch] = a[(int64_t)row * c + ch] * inv;
}

template <typename T>
int clip_similarity_t(const void *feat, int64_t n, int c, const float *anchors, int na, float *sim, float *inv_norm_f,
                      void *workspace, hipStream_t s) {
  constexpr int EPL = Tr<T>::EPL, LD = Tr<T>::LD;
  LGS_REQUIRE(c % EPL == 0, "lgs_clip_similarity: feature dim must be a multiple of the 16-byte load width");
  LGS_REQUIRE(na % 4 == 0, "lgs_clip_similarity: anchor count must be a multiple of 4");
  if (n == 0) return 0;
  const int nc = pad32(c) / 32, nb_total = pad32(na) / 32;
  char *ws = reinterpret_cast<char *>(workspace);
  float *tn = reinterpret_cast<float *>(ws);
  int64_t off = align256((int64_t)na * c * 4);
  uint4 *wp = reinterpret_cast<uint4 *>(ws + off);
  off += align256((int64_t)(nc + 3) * (nb_total + 3) * LD * 64 * 16);
  float *inv = inv_norm_f ? inv_norm_f : reinterpret_cast<float *>(ws + off);
  const T *f = reinterpret_cast<const T *>(feat);
  LGS_KLAUNCH(k_normalize_anchors, na, 64, 0, s, anchors, na, c, tn);
  View v;
  v.n_pad = pad_rows(n); v.n_out = n; v.n_in = n; v.KS = 1; v.K = 1;
  const GatherCfg cfg = gather_cfg<T>(v, nb_total);
  const int ncp = (nc + cfg.sc - 1) / cfg.sc * cfg.sc, nbp = (nb_total + cfg.wb - 1) / cfg.wb * cfg.wb;
  int64_t total = (int64_t)ncp * nbp * LD * 64;
  // T^[a][c] read as w[o = a][g = c]  ("transposed" form of the packer with cin_w = na, cout_w = c)
  LGS_KLAUNCH((k_pack_weights<T>), (unsigned)((total + 255) / 256), 256, 0, s, tn, 1, na, c, 1, 0, c, na, ncp, nbp, wp);
  LGS_KLAUNCH((k_row_invnorm<T>), (unsigned)((n * 64 + 255) / 256), 256, 0, s, f, n, c, inv);
  LGS_HIP(hipGetLastError());
  return launch_gather<T>(v, cfg, f, c, nc, wp, nb_total, ncp, nbp, 1, (T *)nullptr, na, nullptr, s, sim, inv);
}

// Fused CLIP loss forward: one launch of the EPI = 1 instance.  The wave owns whole rows (all anchor columns), so
// the anchor count is limited to 7 column blocks (224); the 200 ScanNet200 anchors use the 7-block tile.
template <typename T>
int clip_loss_forward_t(const void *feat, int64_t n, int c, const float *anchors, int na, const ClipEpi &ce_in, float *anchors_n,
                        float *sim, void *workspace, hipStream_t s) {
  constexpr int EPL = Tr<T>::EPL, LD = Tr<T>::LD;
  constexpr bool kF32 = (sizeof(T) == 4);
  LGS_REQUIRE(c % EPL == 0, "lgs_clip_loss_forward: feature dim must be a multiple of the 16-byte load width");
  LGS_REQUIRE(na % 4 == 0 && na >= 4 && na <= 224, "lgs_clip_loss_forward: anchor count must be a multiple of 4 in [4, 224]");
  LGS_REQUIRE(ce_in.k_neg >= 1 && ce_in.k_neg <= 7, "lgs_clip_loss_forward: 1..7 negatives per row");
  if (n == 0) return 0;
  const int nc = pad32(c) / 32, nb_total = pad32(na) / 32;
  const int ncb = nb_total <= 1 ? 1 : nb_total <= 2 ? 2 : nb_total <= 4 ? 4 : 7;
  const int sc = ncb == 1 ? (kF32 ? 4 : 8) : ncb == 2 ? (kF32 ? 2 : 4) : ncb == 4 ? (kF32 ? 1 : 2) : (kF32 ? 1 : 2);
  const int ncp = (nc + sc - 1) / sc * sc, nbp = ncb;
  char *ws = reinterpret_cast<char *>(workspace);
  uint4 *wp = reinterpret_cast<uint4 *>(ws);
  LGS_KLAUNCH(k_normalize_anchors, na, 64, 0, s, anchors, na, c, anchors_n);
  int64_t total = (int64_t)ncp * nbp * LD * 64;
  // T^[a][c] read as w[o = a][g = c]  ("transposed" form of the packer with cin_w = na, cout_w = c)
  LGS_KLAUNCH((k_pack_weights<T>), (unsigned)((total + 255) / 256), 256, 0, s, anchors_n, 1, na, c, 1, 0, c, na, ncp, nbp, wp);
  View v;
  v.n_pad = pad_rows(n); v.n_out = n; v.n_in = n; v.KS = 1; v.K = 1;
  const uint64_t in_bytes64 = (uint64_t)n * (uint64_t)c * sizeof(T), w_bytes64 = (uint64_t)total * 16;
  LGS_REQUIRE(in_bytes64 < 0xfffff000ull, "lgs_clip_loss_forward: feature tensor of 4 GiB or more");
  ClipEpi ce = ce_in;
  ce.n_anchor = na;
  const T *f = reinterpret_cast<const T *>(feat);
  dim3 grid((unsigned)(v.n_pad / 128), 1);
#define LGS_CLIP(NCB, SC, D)                                                                                              \
  LGS_KLAUNCH((k_conv_gather<T, 1, NCB, 4, 1, SC, D, 1>), grid, dim3(256), 0, s, v, f, c, nc,                      \
                     reinterpret_cast<const u32x4 *>(wp), nb_total, ncp, nbp, (T *)nullptr, na, (const float *)nullptr, sim, \
                     (const float *)nullptr, (unsigned)in_bytes64, (unsigned)w_bytes64, (int64_t)0, ce, BnEpi(), nc, c)
  switch (ncb) {
    case 1: LGS_CLIP(1, (kF32 ? 4 : 8), (kF32 ? 4 : 8)); break;
    case 2: LGS_CLIP(2, (kF32 ? 2 : 4), (kF32 ? 3 : 4)); break;
    case 4: LGS_CLIP(4, (kF32 ? 1 : 2), (kF32 ? 3 : 4)); break;
    default: LGS_CLIP(7, (kF32 ? 1 : 2), (kF32 ? 3 : 4)); break;
  }
#undef LGS_CLIP
  LGS_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int pack_desc_t(const View &v, int K, int cin_w, int cout_w, int transposed_w, int mirror, int g_real, int o_real, int dtype,
                lgs_pack_desc *d) {
  constexpr int EPL = Tr<T>::EPL, LD = Tr<T>::LD;
  memset(d, 0, sizeof(*d));
  if (o_real % 4 != 0 || g_real % EPL != 0 || v.n_pad == 0) return 0;      // scratch / padded-input paths pack internally
  const int nc = pad32(g_real) / 32, nb_total = pad32(o_real) / 32;
  const GatherCfg cfg = gather_cfg<T>(v, nb_total);
  d->ncp = (nc + cfg.sc - 1) / cfg.sc * cfg.sc;
  d->nbp = (nb_total + cfg.wb - 1) / cfg.wb * cfg.wb;
  d->K = K; d->cin_w = cin_w; d->cout_w = cout_w; d->transposed = transposed_w; d->mirror = mirror;
  d->g_real = g_real; d->o_real = o_real; d->dtype = Tr<T>::SPLIT ? kDtF32Split : dtype;
  d->total = (int64_t)K * d->ncp * d->nbp * Tr<T>::WLD * 64;
  d->bytes = d->total * 16;
  return 0;
}

inline bool fp32_split_on() { return tune(T_FP32_SPLIT) != 0; }

}  // namespace lgs

using namespace lgs;

extern "C" {

int64_t lgs_conv_workspace_bytes(const lgs_kmap *km, int cin, int cout, int dtype, int op) {
  if (!km) return -1;
  const int e = esize(dtype);
  // the maps of an empty batch: no launch plan to size (the planners divide by range and lane counts); the entry points return
  // before they touch the workspace
  if (km->fwd.n_pad == 0 && km->bwd.n_pad == 0) return 256;
  if (op == 2) return lgs::wgrad_workspace_bytes(km, cin, cout, dtype);
  int g = op == 0 ? cin : cout, o = op == 0 ? cout : cin;
  // packed weights: fp32 images of the split path hold three bf16 pieces per element (6 instead of 4 bytes)
  int64_t bytes = align256((int64_t)km->K * (pad32(g) + 96) * (pad32(o) + 96) * (dtype == LGS_F32 ? 6 : e));
  int64_t nmax = km->fwd.n_in > km->bwd.n_in ? km->fwd.n_in : km->bwd.n_in;
  if (g % epl(dtype) != 0) bytes += align256(nmax * pad32(g) * e);
  if (o % 4 != 0) bytes += align256(nmax * (int64_t)((o + 3) / 4 * 4) * e) + align256(4 * (int64_t)((o + 3) / 4 * 4)) + 256;   // scratch image + padded bias
  int64_t omax = km->fwd.n_out > km->bwd.n_out ? km->fwd.n_out : km->bwd.n_out;
  bytes += lgs::split_partial_bytes(km->K, omax, o) + lgs::split_partial_bytes(km->K, omax, (o + 3) / 4 * 4);
  return bytes + 256;
}

int lgs_conv_bn_partial_rows(const lgs_kmap *km, int transposed, int cout, int dtype) {
  if (!km) return 0;
  const View &v = transposed ? km->bwd : km->fwd;
  if (dtype == LGS_F32) return bn_partial_rows_t<float>(v, km->K, cout);
  if (dtype == LGS_BF16) return bn_partial_rows_t<bf16_t>(v, km->K, cout);
  return 0;
}

int lgs_conv_pack_desc(const lgs_kmap *km, int op, int transposed, int cin, int cout, int dtype, lgs_pack_desc *out) {
  LGS_REQUIRE(km && out && (op == 0 || op == 1), "lgs_conv_pack_desc: bad argument");
  const View &v = op == 0 ? (transposed ? km->bwd : km->fwd) : (transposed ? km->fwd : km->bwd);
  const int mirror = (op == 1 && km->ks == 3) ? 1 : 0;
  const int g = op == 0 ? cin : cout, o = op == 0 ? cout : cin;
  if (dtype == LGS_F32 && fp32_split_on()) return pack_desc_t<f32s_t>(v, km->K, cin, cout, op, mirror, g, o, dtype, out);
  if (dtype == LGS_F32) return pack_desc_t<float>(v, km->K, cin, cout, op, mirror, g, o, dtype, out);
  if (dtype == LGS_BF16) return pack_desc_t<bf16_t>(v, km->K, cin, cout, op, mirror, g, o, dtype, out);
  LGS_REQUIRE(false, "lgs_conv_pack_desc: unknown dtype");
}

int lgs_pack_weights_batch(const lgs_pack_desc *descs_device, int n, int64_t max_total, void *stream) {
  LGS_REQUIRE(descs_device && n > 0 && max_total > 0, "lgs_pack_weights_batch: bad argument");
  dim3 grid((unsigned)((max_total + 255) / 256), (unsigned)n);
  LGS_KLAUNCH(k_pack_weights_batch, grid, 256, 0, (hipStream_t)stream, descs_device);
  LGS_HIP(hipGetLastError());
  return 0;
}

int lgs_conv_forward(lgs_kmap *km, int transposed, const void *in, int cin, const float *weight, int cout,
                     const float *bias, void *out, int dtype, void *workspace, float *bn_partial, const float *bn_pivot,
                     void *packed, int pack_mode, int in_row_stride, void *stream) {
  LGS_REQUIRE(km && weight && workspace, "lgs_conv_forward: null argument");
  LGS_REQUIRE(!(transposed && km->ks == 3), "transposed 3x3x3 convolution is not part of the model family");
  const View &v = transposed ? km->bwd : km->fwd;
  View vv = v; vv.mirror = 0;
  hipStream_t s = (hipStream_t)stream;
  if (kmap_wait(km, s)) return 1;
  BnEpi bn;
  bn.partial = bn_partial; bn.pivot = bn_pivot;
  LGS_REQUIRE(!bn_partial || lgs_conv_bn_partial_rows(km, transposed, cout, dtype) > 0,
              "lgs_conv_forward: this launch shape produces no BatchNorm statistics (see lgs_conv_bn_partial_rows)");
  if (dtype == LGS_F32 && fp32_split_on()) return conv_gather_op<float, f32s_t>(vv, in, cin, weight, km->K, cin, cout, 0, cout, bias, out, workspace, s, -1, bn_partial ? &bn : nullptr, packed, pack_mode, in_row_stride);
  if (dtype == LGS_F32) return conv_gather_op<float>(vv, in, cin, weight, km->K, cin, cout, 0, cout, bias, out, workspace, s, -1, bn_partial ? &bn : nullptr, packed, pack_mode, in_row_stride);
  if (dtype == LGS_BF16) return conv_gather_op<bf16_t>(vv, in, cin, weight, km->K, cin, cout, 0, cout, bias, out, workspace, s, -1, bn_partial ? &bn : nullptr, packed, pack_mode, in_row_stride);
  LGS_REQUIRE(false, "lgs_conv_forward: unknown dtype");
}

int lgs_conv_dgrad(lgs_kmap *km, int transposed, const void *grad_out, int cout, const float *weight, int cin,
                   void *grad_in, int dtype, void *workspace, void *packed, int pack_mode, void *stream) {
  LGS_REQUIRE(km && weight && workspace, "lgs_conv_dgrad: null argument");
  LGS_REQUIRE(!(transposed && km->ks == 3), "transposed 3x3x3 convolution is not part of the model family");
  const View &v = transposed ? km->fwd : km->bwd;  // the opposite direction of the forward
  View vv = v; vv.mirror = (km->ks == 3) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (kmap_wait(km, s)) return 1;
  if (dtype == LGS_F32 && fp32_split_on()) return conv_gather_op<float, f32s_t>(vv, grad_out, cout, weight, km->K, cin, cout, 1, cin, nullptr, grad_in, workspace, s, -1, nullptr, packed, pack_mode);
  if (dtype == LGS_F32) return conv_gather_op<float>(vv, grad_out, cout, weight, km->K, cin, cout, 1, cin, nullptr, grad_in, workspace, s, -1, nullptr, packed, pack_mode);
  if (dtype == LGS_BF16) return conv_gather_op<bf16_t>(vv, grad_out, cout, weight, km->K, cin, cout, 1, cin, nullptr, grad_in, workspace, s, -1, nullptr, packed, pack_mode);
  LGS_REQUIRE(false, "lgs_conv_dgrad: unknown dtype");
}

// 1 if lgs_conv_dgrad_accumulate adds inside the kernel epilogue for this launch shape (everything but the 2-D blocked wide
// kernel and input widths off the 4-channel grid)
int lgs_conv_dgrad_can_accumulate(const lgs_kmap *km, int transposed, int cin, int cout, int dtype) {
  if (!km || (transposed && km->ks == 3) || cin % 4 != 0 || (dtype != LGS_F32 && dtype != LGS_BF16)) return 0;
  const View &v = transposed ? km->fwd : km->bwd;
  if (v.n_pad == 0) return 0;
  const int nb_total = pad32(cin) / 32;
  const int id = dtype == LGS_F32 ? gather_cfg<float>(v, nb_total).id : gather_cfg<bf16_t>(v, nb_total).id;
  return id == 17 ? 0 : 1;
}

// grad_in += dgrad(grad_out): the sum autograd would form when the convolution's input also feeds a residual branch, taken
// in the epilogue (rounded exactly like "store dgrad, then add"), instead of a separate elementwise pass over [N, cin]
int lgs_conv_dgrad_accumulate(lgs_kmap *km, int transposed, const void *grad_out, int cout, const float *weight, int cin,
                              void *grad_in, int dtype, void *workspace, void *packed, int pack_mode, void *stream) {
  LGS_REQUIRE(km && weight && workspace && grad_in, "lgs_conv_dgrad_accumulate: null argument");
  LGS_REQUIRE(lgs_conv_dgrad_can_accumulate(km, transposed, cin, cout, dtype), "lgs_conv_dgrad_accumulate: this launch shape has no accumulating epilogue (ask lgs_conv_dgrad_can_accumulate first)");
  const View &v = transposed ? km->fwd : km->bwd;
  View vv = v; vv.mirror = (km->ks == 3) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (kmap_wait(km, s)) return 1;
  BnEpi acc; acc.accum = 1;
  if (dtype == LGS_F32 && fp32_split_on()) return conv_gather_op<float, f32s_t>(vv, grad_out, cout, weight, km->K, cin, cout, 1, cin, nullptr, grad_in, workspace, s, -1, &acc, packed, pack_mode);
  if (dtype == LGS_F32) return conv_gather_op<float>(vv, grad_out, cout, weight, km->K, cin, cout, 1, cin, nullptr, grad_in, workspace, s, -1, &acc, packed, pack_mode);
  return conv_gather_op<bf16_t>(vv, grad_out, cout, weight, km->K, cin, cout, 1, cin, nullptr, grad_in, workspace, s, -1, &acc, packed, pack_mode);
}

}  // extern "C"

extern "C" {

int64_t lgs_clip_workspace_bytes(int c, int n_anchor, int dtype) {
  int64_t b = align256((int64_t)n_anchor * c * 4) + align256((int64_t)(pad32(c) + 96) * (pad32(n_anchor) + 96) * esize(dtype));
  return b + 256;
}

int64_t lgs_clip_loss_workspace_bytes(int c, int n_anchor, int dtype) {
  (void)n_anchor;
  return align256((int64_t)(pad32(c) + 256) * 256 * esize(dtype)) + 256;   // packed anchors: <= (nc + 7) chunks x 7 blocks
}

int lgs_clip_loss_forward(const void *feat, int64_t n, int c, const float *anchors, int n_anchor, const int64_t *labels,
                          const int64_t *neg, int k_neg, int64_t ignore_label, float *d_pos, float *d_neg, int64_t *pred,
                          float *inv_norm_f, float *anchors_n, float *sim, int dtype, void *workspace, void *stream) {
  LGS_REQUIRE(feat && anchors && labels && neg && d_pos && d_neg && anchors_n && workspace, "lgs_clip_loss_forward: null argument");
  ClipEpi ce;
  ce.labels = labels; ce.neg = neg; ce.k_neg = k_neg; ce.ignore = ignore_label;
  ce.d_pos = d_pos; ce.d_neg = d_neg; ce.pred = pred; ce.inv_norm = inv_norm_f;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LGS_F32) return clip_loss_forward_t<float>(feat, n, c, anchors, n_anchor, ce, anchors_n, sim, workspace, s);
  if (dtype == LGS_BF16) return clip_loss_forward_t<bf16_t>(feat, n, c, anchors, n_anchor, ce, anchors_n, sim, workspace, s);
  LGS_REQUIRE(false, "lgs_clip_loss_forward: unknown dtype");
}

int lgs_clip_similarity(const void *feat, int64_t n, int c, const float *anchors, int n_anchor, float *sim,
                        float *inv_norm_f, int dtype, void *workspace, void *stream) {
  LGS_REQUIRE(feat && anchors && sim && workspace && inv_norm_f, "lgs_clip_similarity: null argument");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LGS_F32) return clip_similarity_t<float>(feat, n, c, anchors, n_anchor, sim, inv_norm_f, workspace, s);
  if (dtype == LGS_BF16) return clip_similarity_t<bf16_t>(feat, n, c, anchors, n_anchor, sim, inv_norm_f, workspace, s);
  LGS_REQUIRE(false, "lgs_clip_similarity: unknown dtype");
}

}  // extern "C"
