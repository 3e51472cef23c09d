// lgs_wgrad.hip -- weight gradient of the sparse convolution on gfx950.
//
// Replaces the wgrad half of MinkowskiConvolution[Transpose]'s autograd backward
//   /root/reference/models/modules/common.py:195-203,228-236 ; call sites res16unet.py:196-270
//
//   gw[k][ci][co] = sum over pairs (i, o) of kernel offset k:  in[i][ci] * gout[o][co]
//
// It is a GEMM whose reduction axis is the (ragged, gathered) pair list, so neither operand is in MFMA
// fragment order in memory: the fragment wants 8 consecutive PAIRS of one channel per lane while HBM holds
// rows of channels.  bf16 path (k_wgrad_bf16): each wavefront works alone -- it ballot-compacts the valid
// pairs of its position range into LDS, then per group of 16 pairs gathers the 16 input rows and 16
// grad rows with 16-byte loads (whole 64..256-byte row pieces), parks them row-major in a wave-private
// LDS tile and reads them back TRANSPOSED with ds_read_b64_tr_b16 (gfx950), which yields exactly the
// 32x32x16 bf16 MFMA operand layout; the row stride is chosen = 64 (mod 256) bytes so the transposing
// reads are bank-conflict free.  No workgroup barriers.  Partials per wave slot are written once and folded
// by k_wgrad_reduce in a fixed order (deterministic, no float atomics).
// fp32 path (k_wgrad_f32): v_mfma_f32_32x32x2_f32 takes ONE element per lane, so rows are read straight
// from HBM in operand order (32 consecutive channels of 2 pairs per instruction) -- exact fp32.
#include "lgs_common.h"

#include <stdlib.h>
#include <type_traits>

namespace lgs {

// ------------------------------------------------------------------------------------ fp32 (exact) path
constexpr int kWgChunk = 512;  // positions compacted per iteration

template <typename T, int NCB>
__global__ __launch_bounds__(256) void k_wgrad_f32(View v, const T *__restrict__ in, int cin_real, const T *__restrict__ gout,
                                                   int cout_real, int cin_pad, int cout_pad, int64_t span,
                                                   float *__restrict__ partial) {
  __shared__ int32_t l_in[kWgChunk], l_out[kWgChunk];
  __shared__ int32_t l_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 31, h = lane >> 5;
  const int k = blockIdx.y;
  const int n_cot = cout_pad / (32 * NCB);
  const int cot = blockIdx.z % n_cot, cig = blockIdx.z / n_cot;
  const int cib = cig * 4 + wave;
  const bool wave_active = cib * 32 < cin_pad;
  const int slot = v.KS > 1 ? k : 0;

  f32x16 acc[NCB];
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int64_t p_begin = (int64_t)blockIdx.x * span;
  const int64_t p_end = min(p_begin + span, v.n_pad);
  for (int64_t base = p_begin; base < p_end; base += kWgChunk) {
    // ---- compact valid pairs of this chunk (two positions per thread, fixed wave order)
    int32_t my_in[2], my_out[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int64_t p = base + wave * 128 + u * 64 + lane;
      int32_t i = -1, o = -1;
      if (p < p_end) {
        bool grp_ok = true;
        if (v.KS > 1) grp_ok = (v.mask64[p >> 6] >> slot) & 1u;
        else if (v.tile_k) grp_ok = v.tile_k[p >> 6] == k;
        if (grp_ok) {
          o = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
          i = v.nbr ? v.nbr[(int64_t)slot * v.n_pad + p] : (p < v.n_in ? (int32_t)p : -1);
        }
      }
      my_in[u] = (i >= 0 && o >= 0) ? i : -1;
      my_out[u] = o;
    }
    unsigned long long bal0 = __ballot(my_in[0] >= 0), bal1 = __ballot(my_in[1] >= 0);
    int c0 = (int)__builtin_popcountll(bal0), c1 = (int)__builtin_popcountll(bal1);
    if (lane == 0) l_cnt[wave] = c0 + c1;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int c = l_cnt[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (my_in[0] >= 0) {
      int at = wbase + (int)__builtin_popcountll(bal0 & ((1ull << lane) - 1ull));
      l_in[at] = my_in[0]; l_out[at] = my_out[0];
    }
    if (my_in[1] >= 0) {
      int at = wbase + c0 + (int)__builtin_popcountll(bal1 & ((1ull << lane) - 1ull));
      l_in[at] = my_in[1]; l_out[at] = my_out[1];
    }
    __syncthreads();
    // ---- MFMA over the compacted pairs, two pairs (k = h) per 32x32x2 instruction
    if (wave_active) {
      const int ci = cib * 32 + vx;
      for (int j = 0; j < total; j += 8) {
        float a[4], b[4][NCB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int pr = j + 2 * u + h;
          bool ok = pr < total;
          int32_t irow = ok ? l_in[pr] : 0, orow = ok ? l_out[pr] : 0;
          a[u] = (ok && ci < cin_real) ? ld_elem(in + (int64_t)irow * cin_real + ci) : 0.f;
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb) {
            int co = (cot * NCB + nb) * 32 + vx;
            b[u][nb] = (ok && co < cout_real) ? ld_elem(gout + (int64_t)orow * cout_real + co) : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][nb], acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (!wave_active) return;
  // D[i = ci][j = co]: lane holds column j = vx, rows (r&3) + 8*(r>>2) + 4*h
  float *dst = partial + (((int64_t)blockIdx.x * v.K + k) * cin_pad) * cout_pad;
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb) {
    int co = (cot * NCB + nb) * 32 + vx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int ci = cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[(int64_t)ci * cout_pad + co] = acc[nb][r];
    }
  }
}

// fp32 rows staged through LDS (round 5).  k_wgrad_f32 above feeds v_mfma_f32_32x32x2_f32 with one 4-byte load per lane and
// operand: 16 wave-wide load instructions per 12 MFMAs and wave, ~6 per pair at 96 x 96 channels -- the fp32 training step spent
// 75 ms per step in it (profiles/r05_bench.json, fp32.roofline.discovery_step.wgrad), all of it on the CU's vector-memory
// instruction rate.  Here the workgroup brings the rows of 32 compacted pairs into LDS with 16-byte loads (input rows: the 128
// channels of the workgroup's four waves, gradient rows: the NCB x 32 channels of its column tile; every byte once per workgroup
// instead of once per wave), the next 32 pairs are in flight in registers while the current ones are multiplied, and the MFMA
// operands are 4-byte LDS reads (row pitch + 32 floats: the two pair rows of one instruction fall on different banks).  The
// MFMA sequence -- pairs in compacted order, two per instruction -- is k_wgrad_f32's, so the partial slabs are bit-identical.
// Needs rows on the 16-byte grid (the 3-channel input layer and odd head widths keep k_wgrad_f32).
template <int NCB>
__global__ __launch_bounds__(256) void k_wgrad_f32_lds(View v, const float *__restrict__ in, int cin_real, const float *__restrict__ gout,
                                                       int cout_real, int cin_pad, int cout_pad, int64_t span,
                                                       float *__restrict__ partial) {
  constexpr int PB = 16;                       // pairs per staged sub-chunk (16 / 32 / 64 at level 0, 96 x 96: 3.1 / 3.5 / 3.7 ms)
  constexpr int AS = 128 + 32, BS = NCB * 32 + 32;
  constexpr int NB4 = NCB * 8;                 // float4 pieces per gradient row
  __shared__ int32_t l_in[kWgChunk], l_out[kWgChunk];
  __shared__ int32_t l_cnt[4];
  __shared__ __attribute__((aligned(16))) float sA[PB * AS];
  __shared__ __attribute__((aligned(16))) float sB[PB * BS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 31, h = lane >> 5;
  // 3^3 maps: the centre offset pairs every position, a corner offset ~15 % of them -- the heavy offsets are dispatched first
  const int y = blockIdx.y;
  const int k = v.K == 27 ? (y == 0 ? 13 : (y & 1) ? 13 - (y + 1) / 2 : 13 + y / 2) : y;
  const int n_cot = cout_pad / (32 * NCB);
  const int cot = blockIdx.z % n_cot, cig = blockIdx.z / n_cot;
  const int cib = cig * 4 + wave;
  const bool wave_active = cib * 32 < cin_pad;
  const int slot = v.KS > 1 ? k : 0;
  const int ca0 = cig * 128, cb0 = cot * NCB * 32;

  f32x16 acc[NCB];
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  constexpr int NA = PB * 32 / 256, NBQ = (PB * NB4 + 255) / 256;
  float4 ra[NA], rb[NBQ];
  auto fetch = [&](int sub, int total) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, r = idx >> 5, c4 = idx & 31;
      const int pr = sub * PB + r, ch = ca0 + 4 * c4;
      ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pr < total && ch < cin_real) ra[u] = *reinterpret_cast<const float4 *>(in + (int64_t)l_in[pr] * cin_real + ch);
    }
#pragma unroll
    for (int u = 0; u < NBQ; ++u) {
      const int idx = tid + 256 * u, r = idx / NB4, c4 = idx % NB4;
      const int pr = sub * PB + r, ch = cb0 + 4 * c4;
      rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < PB * NB4 && pr < total && ch < cout_real) rb[u] = *reinterpret_cast<const float4 *>(gout + (int64_t)l_out[pr] * cout_real + ch);
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, r = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4 *>(sA + r * AS + 4 * c4) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < NBQ; ++u) {
      const int idx = tid + 256 * u, r = idx / NB4, c4 = idx % NB4;
      if (idx < PB * NB4) *reinterpret_cast<float4 *>(sB + r * BS + 4 * c4) = rb[u];
    }
  };

  const int64_t p_begin = (int64_t)blockIdx.x * span;
  const int64_t p_end = min(p_begin + span, v.n_pad);
  for (int64_t base = p_begin; base < p_end; base += kWgChunk) {
    // ---- compact valid pairs of this chunk (two positions per thread, fixed wave order): as k_wgrad_f32
    int32_t my_in[2], my_out[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int64_t p = base + wave * 128 + u * 64 + lane;
      int32_t i = -1, o = -1;
      if (p < p_end) {
        bool grp_ok = true;
        if (v.KS > 1) grp_ok = (v.mask64[p >> 6] >> slot) & 1u;
        else if (v.tile_k) grp_ok = v.tile_k[p >> 6] == k;
        if (grp_ok) {
          o = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
          i = v.nbr ? v.nbr[(int64_t)slot * v.n_pad + p] : (p < v.n_in ? (int32_t)p : -1);
        }
      }
      my_in[u] = (i >= 0 && o >= 0) ? i : -1;
      my_out[u] = o;
    }
    unsigned long long bal0 = __ballot(my_in[0] >= 0), bal1 = __ballot(my_in[1] >= 0);
    int c0 = (int)__builtin_popcountll(bal0), c1 = (int)__builtin_popcountll(bal1);
    if (lane == 0) l_cnt[wave] = c0 + c1;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int c = l_cnt[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (my_in[0] >= 0) {
      int at = wbase + (int)__builtin_popcountll(bal0 & ((1ull << lane) - 1ull));
      l_in[at] = my_in[0]; l_out[at] = my_out[0];
    }
    if (my_in[1] >= 0) {
      int at = wbase + c0 + (int)__builtin_popcountll(bal1 & ((1ull << lane) - 1ull));
      l_in[at] = my_in[1]; l_out[at] = my_out[1];
    }
    __syncthreads();
    const int nsub = (total + PB - 1) / PB;
    if (nsub > 0) fetch(0, total);
    for (int sub = 0; sub < nsub; ++sub) {
      stage();
      __syncthreads();
      if (sub + 1 < nsub) fetch(sub + 1, total);             // in flight while this sub-chunk is multiplied
      if (wave_active) {
        const int np = min(PB, (total - sub * PB + 7) & ~7);     // rows behind `total` are staged as zeros: whole groups of 8 pairs
        const float *pa = sA + h * AS + wave * 32 + vx;
        const float *pb = sB + h * BS + vx;
        for (int p8 = 0; p8 < np; p8 += 8) {
          float a[4], b[4][NCB];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            a[q] = pa[(p8 + 2 * q) * AS];
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) b[q][nb] = pb[(p8 + 2 * q) * BS + nb * 32];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nb = 0; nb < NCB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q][nb], acc[nb], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
  if (!wave_active) return;
  float *dst = partial + (((int64_t)blockIdx.x * v.K + k) * cin_pad) * cout_pad;
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb) {
    int co = (cot * NCB + nb) * 32 + vx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int ci = cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[(int64_t)ci * cout_pad + co] = acc[nb][r];
    }
  }
}

// ------------------------------------------------------------------------------------ bf16 MFMA path
constexpr int kQ = 1024;       // positions compacted per wave chunk
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// row stride (bytes) of a [16 pairs][C channels] bf16 tile, = 64 (mod 256) -> conflict-free tr reads
__host__ __device__ constexpr int tile_stride(int c) {
  int s = c * 2;
  while ((s / 64) % 4 != 1 && (s / 64) % 4 != 3) s += 64;
  return s;
}

typedef short short4v __attribute__((ext_vector_type(4)));

// fp32 rows, bf16 products (round 5, knob WGRAD_F32_LDS = 2): k_wgrad_f32_lds's structure -- compacted pairs, rows staged in LDS
// by the whole workgroup, the next 16 pairs in flight during the multiply -- with every fp32 element split exactly into three bf16
// pieces (x = hi + mid + lo by truncation, as in k_conv_gather's Tr<f32s_t>) WHEN IT IS STAGED, once per workgroup: three
// row-major bf16 planes per operand, read back transposed (ds_read_b64_tr_b16, the idiom of k_wgrad_bf16) as 32x32x16 operands.
// Per 16 pairs and 32 x 32 tile: six bf16 MFMAs (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; 6 x 32 cycles) instead of eight
// v_mfma_f32_32x32x2_f32 (8 x 64 cycles); dropped terms < 2^-24 |x g|, fp32 accumulation.
__device__ inline void split3_bf16(float x, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  const uint32_t h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);                  // exact
  const uint32_t m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);                 // exact
  hi = h >> 16; mid = m >> 16; lo = __float_as_uint(r2) >> 16;
}
__device__ inline void split3_store(char *plane0, int plane_bytes, int off, const float4 &v4) {
  uint32_t h[4], m[4], l[4];
  split3_bf16(v4.x, h[0], m[0], l[0]); split3_bf16(v4.y, h[1], m[1], l[1]);
  split3_bf16(v4.z, h[2], m[2], l[2]); split3_bf16(v4.w, h[3], m[3], l[3]);
  *reinterpret_cast<uint2 *>(plane0 + off) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  *reinterpret_cast<uint2 *>(plane0 + plane_bytes + off) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
  *reinterpret_cast<uint2 *>(plane0 + 2 * plane_bytes + off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}
__device__ inline bf16x8 tr_operand(const char *p0, int row_stride) {
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(p0));
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(p0 + 4 * row_stride));
  u32x4 pk;
  pk.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  pk.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  pk.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  pk.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return __builtin_bit_cast(bf16x8, pk);
}

template <int NCB>
__global__ __launch_bounds__(256) void k_wgrad_f32s_lds(View v, const float *__restrict__ in, int cin_real, const float *__restrict__ gout,
                                                        int cout_real, int cin_pad, int cout_pad, int64_t span,
                                                        float *__restrict__ partial) {
  constexpr int PB = 16;                                   // one 32x32x16 k-group of pairs per staged sub-chunk
  constexpr int SA = tile_stride(128), SG = tile_stride(NCB * 32);   // row pitch (bytes) of the bf16 planes
  constexpr int PA = PB * SA, PG = PB * SG;                // bytes per plane
  constexpr int NB4 = NCB * 8;
  constexpr int NA = PB * 32 / 256, NBQ = (PB * NB4 + 255) / 256;
  __shared__ int32_t l_in[kWgChunk], l_out[kWgChunk];
  __shared__ int32_t l_cnt[4];
  __shared__ __attribute__((aligned(16))) char sA[3 * PA];
  __shared__ __attribute__((aligned(16))) char sB[3 * PG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vx = lane & 31, h = lane >> 5;
  const int y = blockIdx.y;
  const int k = v.K == 27 ? (y == 0 ? 13 : (y & 1) ? 13 - (y + 1) / 2 : 13 + y / 2) : y;     // heavy offsets first (k_wgrad_f32_lds)
  const int n_cot = cout_pad / (32 * NCB);
  const int cot = blockIdx.z % n_cot, cig = blockIdx.z / n_cot;
  const int cib = cig * 4 + wave;
  const bool wave_active = cib * 32 < cin_pad;
  const int slot = v.KS > 1 ? k : 0;
  const int ca0 = cig * 128, cb0 = cot * NCB * 32;
  // transposing read: 16-lane group g = lane >> 4: cb = g & 1 (16-channel half), pairs 8 (g >> 1) ..; lane i = lane & 15 supplies
  // the 8-byte address (row 8 (g >> 1) + i / 4 [+ 4], channel 16 cb + 4 (i % 4)) -- see k_wgrad_bf16
  const int g16 = lane >> 4, i16 = lane & 15;
  const int tr_row = 8 * (g16 >> 1) + (i16 >> 2), tr_col = 16 * (g16 & 1) + 4 * (i16 & 3);
  const char *pa = sA + tr_row * SA + (32 * wave + tr_col) * 2;
  const char *pg = sB + tr_row * SG + tr_col * 2;

  f32x16 acc[NCB];
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  float4 ra[NA], rb[NBQ];
  auto fetch = [&](int sub, int total) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, r = idx >> 5, c4 = idx & 31;
      const int pr = sub * PB + r, ch = ca0 + 4 * c4;
      ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pr < total && ch < cin_real) ra[u] = *reinterpret_cast<const float4 *>(in + (int64_t)l_in[pr] * cin_real + ch);
    }
#pragma unroll
    for (int u = 0; u < NBQ; ++u) {
      const int idx = tid + 256 * u, r = idx / NB4, c4 = idx % NB4;
      const int pr = sub * PB + r, ch = cb0 + 4 * c4;
      rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < PB * NB4 && pr < total && ch < cout_real) rb[u] = *reinterpret_cast<const float4 *>(gout + (int64_t)l_out[pr] * cout_real + ch);
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u, r = idx >> 5, c4 = idx & 31;
      split3_store(sA, PA, r * SA + 8 * c4, ra[u]);
    }
#pragma unroll
    for (int u = 0; u < NBQ; ++u) {
      const int idx = tid + 256 * u, r = idx / NB4, c4 = idx % NB4;
      if (idx < PB * NB4) split3_store(sB, PG, r * SG + 8 * c4, rb[u]);
    }
  };

  const int64_t p_begin = (int64_t)blockIdx.x * span;
  const int64_t p_end = min(p_begin + span, v.n_pad);
  for (int64_t base = p_begin; base < p_end; base += kWgChunk) {
    int32_t my_in[2], my_out[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int64_t p = base + wave * 128 + u * 64 + lane;
      int32_t i = -1, o = -1;
      if (p < p_end) {
        bool grp_ok = true;
        if (v.KS > 1) grp_ok = (v.mask64[p >> 6] >> slot) & 1u;
        else if (v.tile_k) grp_ok = v.tile_k[p >> 6] == k;
        if (grp_ok) {
          o = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
          i = v.nbr ? v.nbr[(int64_t)slot * v.n_pad + p] : (p < v.n_in ? (int32_t)p : -1);
        }
      }
      my_in[u] = (i >= 0 && o >= 0) ? i : -1;
      my_out[u] = o;
    }
    unsigned long long bal0 = __ballot(my_in[0] >= 0), bal1 = __ballot(my_in[1] >= 0);
    int c0 = (int)__builtin_popcountll(bal0), c1 = (int)__builtin_popcountll(bal1);
    if (lane == 0) l_cnt[wave] = c0 + c1;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int c = l_cnt[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (my_in[0] >= 0) {
      int at = wbase + (int)__builtin_popcountll(bal0 & ((1ull << lane) - 1ull));
      l_in[at] = my_in[0]; l_out[at] = my_out[0];
    }
    if (my_in[1] >= 0) {
      int at = wbase + c0 + (int)__builtin_popcountll(bal1 & ((1ull << lane) - 1ull));
      l_in[at] = my_in[1]; l_out[at] = my_out[1];
    }
    __syncthreads();
    const int nsub = (total + PB - 1) / PB;
    if (nsub > 0) fetch(0, total);
    for (int sub = 0; sub < nsub; ++sub) {
      stage();
      __syncthreads();
      if (sub + 1 < nsub) fetch(sub + 1, total);
      if (wave_active) {
        bf16x8 fa[3], fg[3][NCB];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[p] = tr_operand(pa + p * PA, SA);
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb) fg[p][nb] = tr_operand(pg + p * PG + 64 * nb, SG);
        }
#pragma unroll
        for (int nb = 0; nb < NCB; ++nb) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fg[1][nb], acc[nb], 0, 0, 0);   // small terms first
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fg[0][nb], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fg[2][nb], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fg[0][nb], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fg[1][nb], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fg[0][nb], acc[nb], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
  if (!wave_active) return;
  float *dst = partial + (((int64_t)blockIdx.x * v.K + k) * cin_pad) * cout_pad;
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb) {
    int co = (cot * NCB + nb) * 32 + vx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int ci = cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[(int64_t)ci * cout_pad + co] = acc[nb][r];
    }
  }
}

// Grid: 1-D.  Workgroup L runs on XCD L % 8 (dispatcher behaviour, used for speed only).  A workgroup owns ONE
// position range (a few thousand Morton-consecutive voxels: its rows + halo fit the XCD's L2) and its four waves
// take four different kernel offsets k of that range (or four sub-ranges when K = 1); the offset groups of a range
// are consecutive in dispatch order on the same XCD.  Every row of the range is therefore fetched from HBM about
// once and re-read ~K times from L1/L2 -- without this ordering the gathers were 98 % L2 misses (PMC) and the
// kernel ran at the random-access rate of HBM.
template <int NCI, int NCO, int D, int OCC>
__global__ __launch_bounds__(256, OCC) void k_wgrad_bf16(View v, const bf16_t *__restrict__ in, int cin_real,
                                                    const bf16_t *__restrict__ gout, int cout_real, int cin_pad,
                                                    int cout_pad, int64_t range, int kpw, int n_ci_tasks, int n_tasks,
                                                    int n_ranges, int n_lanes, float *__restrict__ partial,
                                                    unsigned in_bytes, unsigned gout_bytes) {
  constexpr int CA = 32 * NCI, CG = 32 * NCO;
  constexpr int SA = tile_stride(CA), SG = tile_stride(CG);
  constexpr int LA = (16 * CA / 8 + 63) / 64, LG = (16 * CG / 8 + 63) / 64;  // 16-byte loads per lane per group
  constexpr int WAVE_BYTES = 2 * kQ * 4 + 16 * SA + 16 * SG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char *wbase = smem + wave * WAVE_BYTES;
  int32_t *q_in = reinterpret_cast<int32_t *>(wbase);
  int32_t *q_out = q_in + kQ;
  char *stA = wbase + 2 * kQ * 4;
  char *stG = stA + 16 * SA;

  const int K = v.K;
  const int KG = (K + kpw - 1) / kpw;   // offset groups per range
  const int wpk = 4 / kpw;              // waves sharing one offset (sub-ranges)
  const unsigned L = blockIdx.x, xcd = L & 7u, t = L >> 3;
  const int kg = (int)(t % (unsigned)KG);
  const int task = (int)((t / (unsigned)KG) % (unsigned)n_tasks);
  const int lane_i = (int)(t / (unsigned)(KG * n_tasks)) * 8 + (int)xcd;   // "range lane": ranges lane_i, lane_i + n_lanes, ...
  if (lane_i >= n_lanes) return;
  const int k = kg * kpw + wave / wpk;   // neighbouring offsets share a workgroup: their gathers overlap in L1
  const int sub = wave % wpk;
  if (k >= K) return;  // wave-uniform; no workgroup barriers in this kernel
  const int cit = task % n_ci_tasks, cot = task / n_ci_tasks;
  const int ci0 = cit * CA, co0 = cot * CG;
  const int slot = v.KS > 1 ? k : 0;
  const int64_t wslot = lane_i;         // one partial slab per (lane, offset): sub-range waves are folded in LDS below
  const int64_t sub_len = range / wpk;

  f32x16 acc[NCI][NCO];
#pragma unroll
  for (int a = 0; a < NCI; ++a)
#pragma unroll
    for (int b = 0; b < NCO; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // tr read: 16-lane group g = lane>>4: cb = g&1 (16-channel half), h = g>>1 (pairs 8h..8h+7); lane i = lane&15
  // supplies the 8-byte address (row 8h + 4*rd + i/4, channel 32*blk + 16*cb + 4*(i%4))
  const int g16 = lane >> 4, i16 = lane & 15;
  const int tr_row = 8 * (g16 >> 1) + (i16 >> 2);
  const int tr_col = 16 * (g16 & 1) + 4 * (i16 & 3);

  u32x4 ra[D][LA], rg[D][LG];
  // every lane issues exactly LA + LG 16-byte BUFFER loads per group; a missing row or a padded channel gets an
  // out-of-range offset and the hardware returns zeros (no branch, no 64-bit address math, no zero page)
  constexpr unsigned kOOB = 0xfffff000u;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(gout), 0, (int)gout_bytes, 0x00020000);
  const unsigned arow_b = (unsigned)cin_real * 2u, grow_b = (unsigned)cout_real * 2u;
  unsigned a_ch[LA], g_ch[LG];   // loop-invariant per-lane channel byte offsets (kOOB when the piece is padding)
#pragma unroll
  for (int q = 0; q < LA; ++q) {
    const int id = q * 64 + lane;
    const int ch = (id % (CA / 8)) * 8;
    a_ch[q] = (id < 16 * (CA / 8) && ci0 + ch + 8 <= cin_real) ? (unsigned)(ci0 + ch) * 2u : kOOB;
  }
#pragma unroll
  for (int q = 0; q < LG; ++q) {
    const int id = q * 64 + lane;
    const int ch = (id % (CG / 8)) * 8;
    g_ch[q] = (id < 16 * (CG / 8) && co0 + ch + 8 <= cout_real) ? (unsigned)(co0 + ch) * 2u : kOOB;
  }
  auto issue = [&](int g, u32x4 (&xa)[LA], u32x4 (&xg2)[LG]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < LA; ++q) {
      const int row = ((q * 64 + lane) / (CA / 8)) & 15;
      const int32_t r = q_in[g * 16 + row];
      const unsigned off = (r >= 0 && a_ch[q] != kOOB) ? (unsigned)r * arow_b + a_ch[q] : kOOB;
      xa[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, off, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < LG; ++q) {
      const int row = ((q * 64 + lane) / (CG / 8)) & 15;
      const int32_t r = q_out[g * 16 + row];
      const unsigned off = (r >= 0 && g_ch[q] != kOOB) ? (unsigned)r * grow_b + g_ch[q] : kOOB;
      xg2[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off, 0, 0);
    }
  };
  auto consume = [&](const u32x4 (&xa)[LA], const u32x4 (&xg2)[LG]) __attribute__((always_inline)) {
    // park the 16 x CA and 16 x CG row pieces row-major in the wave's LDS tile
#pragma unroll
    for (int q = 0; q < LA; ++q) {
      const int id = q * 64 + lane;
      const int row = id / (CA / 8), ch = (id % (CA / 8)) * 8;
      if (row < 16) *reinterpret_cast<u32x4 *>(stA + row * SA + ch * 2) = xa[q];
    }
#pragma unroll
    for (int q = 0; q < LG; ++q) {
      const int id = q * 64 + lane;
      const int row = id / (CG / 8), ch = (id % (CG / 8)) * 8;
      if (row < 16) *reinterpret_cast<u32x4 *>(stG + row * SG + ch * 2) = xg2[q];
    }
    __builtin_amdgcn_wave_barrier();
    // transposing reads -> MFMA operands (lane: channel = lane&31 of the block, pairs 8h..8h+7)
    bf16x8 fa[NCI], fg[NCO];
#pragma unroll
    for (int a = 0; a < NCI; ++a) {
      const char *p0 = stA + tr_row * SA + (32 * a + tr_col) * 2;
      short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(p0));
      short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(p0 + 4 * SA));
      u32x4 pk;
      pk.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
      pk.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
      pk.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
      pk.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      fa[a] = __builtin_bit_cast(bf16x8, pk);
    }
#pragma unroll
    for (int b = 0; b < NCO; ++b) {
      const char *p0 = stG + tr_row * SG + (32 * b + tr_col) * 2;
      short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(p0));
      short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(p0 + 4 * SG));
      u32x4 pk;
      pk.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
      pk.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
      pk.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
      pk.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      fg[b] = __builtin_bit_cast(bf16x8, pk);
    }
#pragma unroll
    for (int a = 0; a < NCI; ++a)
#pragma unroll
      for (int b = 0; b < NCO; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fg[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
  };

  // The grid is sized to the resident workgroup slots of the chip and every workgroup walks its lane's ranges with
  // the accumulators kept in registers: one partial slab per LANE instead of one per range (4x fewer partial bytes
  // to write and re-read at level 0), and the KG workgroups that share a range's rows are co-resident by construction.
  for (int rg_i = lane_i; rg_i < n_ranges; rg_i += n_lanes) {
  const int64_t p_begin = (int64_t)rg_i * range + sub * sub_len;
  const int64_t p_end = min(p_begin + sub_len, v.n_pad);
  for (int64_t base = p_begin; base < p_end; base += kQ) {
    // ---- ballot-compact the valid pairs of up to kQ positions (wave-private, in order).  All index loads of the
    // chunk are issued first (16 independent loads in flight instead of 16 dependent round trips), then compacted.
    int n = 0;
    constexpr int NIT = kQ / 64;
    int32_t iv[NIT], ov[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int64_t b = base + u * 64;
      const bool in_rng = b < p_end;                      // wave-uniform (ranges are multiples of 64)
      const int64_t bb = in_rng ? b : p_begin;            // clamp: loads stay in bounds, result masked below
      bool grp_ok = in_rng;
      if (v.KS > 1) grp_ok = grp_ok && ((v.mask64[bb >> 6] >> slot) & 1u);
      else if (v.tile_k) grp_ok = grp_ok && (v.tile_k[bb >> 6] == k);
      const int64_t p = bb + lane;
      int32_t o = -1, i = -1;
      if (grp_ok) {   // wave-uniform: groups whose mask lacks this offset (about half of them) cost no index loads
        o = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
        i = v.nbr ? v.nbr[(int64_t)slot * v.n_pad + p] : (p < v.n_in ? (int32_t)p : -1);
      }
      iv[u] = i;
      ov[u] = o;
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const bool ok = (iv[u] >= 0) && (ov[u] >= 0);
      const unsigned long long bal = __ballot(ok);
      if (ok) {
        const int at = n + (int)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
        q_in[at] = iv[u]; q_out[at] = ov[u];
      }
      n += (int)__builtin_popcountll(bal);
    }
    const int ngroups = (n + 15) >> 4;
    if (lane < 16 && n + lane < ngroups * 16) { q_in[n + lane] = -1; q_out[n + lane] = -1; }
    __builtin_amdgcn_wave_barrier();

    // ---- D-deep ring: the rows of group g+D-1 are in flight while group g goes through LDS and the MFMAs
    int issued = 0, done = 0;
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (issued < ngroups) { issue(issued, ra[d], rg[d]); ++issued; }
    while (ngroups - issued >= D) {   // steady state: unconditional issue -> counted vmcnt waits
#pragma unroll
      for (int d = 0; d < D; ++d) {
        issue(issued + d, ra[(d + D - 1) % D], rg[(d + D - 1) % D]);
        consume(ra[d], rg[d]);
      }
      issued += D;
      done += D;
    }
    while (done < ngroups) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (done < ngroups) {
          if (issued < ngroups) { issue(issued, ra[(d + D - 1) % D], rg[(d + D - 1) % D]); ++issued; }
          consume(ra[d], rg[d]);
          ++done;
        }
      }
    }
  }
  }
  // ---- waves that split one offset's range (wpk > 1: all waves of the workgroup are alive, same k) fold their
  // accumulators into wave 0 through the now idle staging LDS, in a fixed order
  if (wpk > 1) {
    float *red = reinterpret_cast<float *>(smem);
    for (int rr = 1; rr < wpk; ++rr) {
      __syncthreads();
      if (sub == rr) {
#pragma unroll
        for (int a = 0; a < NCI; ++a)
#pragma unroll
          for (int b = 0; b < NCO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((a * NCO + b) * 16 + r) * 64 + lane] = acc[a][b][r];
      }
      __syncthreads();
      if (sub == 0) {
#pragma unroll
        for (int a = 0; a < NCI; ++a)
#pragma unroll
          for (int b = 0; b < NCO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * NCO + b) * 16 + r) * 64 + lane];
      }
    }
    if (sub != 0) return;
  }
  // ---- partial[wslot][k][ci][co]: D[i = ci][j = co], lane holds column j = lane&31, rows (r&3)+8(r>>2)+4h
  const int vx = lane & 31, h = lane >> 5;
  float *dst = partial + ((wslot * v.K + k) * cin_pad) * (int64_t)cout_pad;
#pragma unroll
  for (int a = 0; a < NCI; ++a)
#pragma unroll
    for (int b = 0; b < NCO; ++b) {
      const int co = co0 + 32 * b + vx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (ci < cin_pad && co < cout_pad) dst[(int64_t)ci * cout_pad + co] = acc[a][b][r];
      }
    }
}

// gw[k][ci][co] = sum over the S partial slabs, fixed order.  One thread = 4 consecutive output channels (16-byte
// loads, the padded slab rows are 128-byte aligned), four slabs in flight: the reduce reads S x K x Cin x Cout floats
// (295 MB at level 0) and used to run as one dependent 4-byte load per slab and thread.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int S, int K, int cin_pad, int cout_pad,
                                                      int cin, int cout, float *__restrict__ gw) {
  const int cq = (cout + 3) / 4;                       // channel quads per row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * cin * cq;
  if (idx >= total) return;
  const int q = (int)(idx % cq);
  const int ci = (int)((idx / cq) % cin);
  const int k = (int)(idx / ((int64_t)cq * cin));
  const int64_t slab = (int64_t)K * cin_pad * cout_pad;
  const float *src = partial + ((int64_t)k * cin_pad + ci) * cout_pad + 4 * q;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int x = 0;
  for (; x + 4 <= S; x += 4) {
    const float4 v0 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 0) * slab);
    const float4 v1 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 1) * slab);
    const float4 v2 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 2) * slab);
    const float4 v3 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 3) * slab);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; x < S; ++x) {
    const float4 v0 = *reinterpret_cast<const float4 *>(src + (int64_t)x * slab);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
  }
  const float r[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                      (a0.w + a1.w) + (a2.w + a3.w)};
  float *dst = gw + ((int64_t)k * cin + ci) * cout + 4 * q;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * q + j < cout) dst[j] = r[j];
}

// ------------------------------------------------------------------------------------ position-stationary bf16 path
// k_wgrad_ps (round 2).  The pair-list kernel above reads BOTH rows of every pair (in[i] and gout[o]) from L1/L2 --
// the gradient row of a voxel ~11 times -- and leaves one fp32 partial slab per (range, offset) workgroup.  Here the
// positions are stationary: ONE 512-thread workgroup per CU walks chunks of 128 consecutive map positions and keeps the
// accumulators of ALL K offsets for a 32-channel slice of the gathered operand in the registers of its 8 waves
// (wave w owns 3-4 offsets, balanced by expected pair density).  Per chunk
//   * the STATIONARY operand's rows (gout of the 128 positions; `in` for a transposed conv) are staged ONCE in LDS and
//     serve every offset of every wave: each wave tr-reads them as MFMA B operands once per 16 positions;
//   * per offset only the gathered 64-byte row pieces move: 16 rows x 64 B = one 1 KB tile per (offset, 16 positions),
//     read back transposed (ds_read_b64_tr_b16) as the A operand;
//   * 16-position groups in which an offset has no neighbour at all (wavefront ballot) skip their LDS / MFMA work.
// All data movement is LDS-DMA (buffer_load ... lds): kernel-map indices two chunks ahead, the wave's 16 rows of the
// next stationary tile and the gathered tiles land in LDS without passing through registers (the 192 accumulator
// registers leave no room for a register ring deep enough to cover the gather latency: a 3-deep ring ran at
// 2.3 TB/s).  The schedule is static -- every wave issues the same number of DMA instructions per step, missing
// neighbours are out-of-range offsets that return zeros without memory traffic -- so every wait is a compile-time
// s_waitcnt vmcnt(N).  hipcc would fence each ds_read behind a pending LDS-DMA with vmcnt(0), so the LDS reads of the
// loop are inline asm and the waits are explicit; the per-chunk barrier is a raw s_barrier.
// Partial slabs: one per workgroup "lane" (<= 256 / slices), written once, folded by k_wgrad_reduce_ps in fixed order.
// Measured dead ends: 64-position chunks with a 4-slot ring and three groups in flight (0.65 vs 0.63 ms at L0 96->96:
// the extra barriers cost what the deeper queue gains); skipping neighbour-less tiles with a run-time vmcnt (0.67 ms).
constexpr int kPsChunk = 128;
// offsets per wave for K = 27 (k = (dx+1) + 3(dy+1) + 9(dz+1)): centre / faces / edges / corners spread so that every
// wave sees about the same number of pairs; waves 4..6 own four offsets, the others three
__constant__ int8_t kPsOff27[8][4] = {{13, 0, 2, -1},  {4, 10, 6, -1},  {12, 14, 8, -1},  {16, 22, 18, -1},
                                      {1, 3, 5, 20},   {7, 9, 11, 24},  {15, 17, 19, 26}, {21, 23, 25, -1}};

struct PsArgs {
  View v;
  const bf16_t *G;      // gathered operand  [rows][cg_real]  (rows addressed through v.nbr)
  const bf16_t *S;      // stationary operand [rows][cs_real] (rows addressed through v.out_row / identity)
  int cg_real, cs_real, cg_pad, cs_pad, n_cg, n_cs;
  int g_ld, s_ld;       // row strides (elements) of the two operands: > channels for a column slice of a wider buffer
  int n_lanes, chunks_per_lane, n_chunks, xcd_map;
  float *partial;       // [n_lanes][K][cg_pad][cs_pad]
  unsigned g_bytes, s_bytes, nbr_bytes, orow_bytes;
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define LGS_AS3(p) ((__attribute__((address_space(3))) void *)(p))
// s_waitcnt immediates (gfx9 encoding): vmcnt = N with expcnt / lgkmcnt left alone
#define LGS_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14)))

__host__ __device__ constexpr int ps_wave_lds(int nw) { return 3 * nw * 1024 + 2 * (nw * 512 + 64); }

// the loop of one wave that owns NW kernel offsets (koff) of the map; NCS = 32-channel blocks of the stationary slice
template <int NW, int NCS>
__device__ __forceinline__ void ps_wave(const PsArgs &a, char *smem, const int wave_off, const int (&koff)[NW], const int wlane,
                                        const int cg0, const int cs0, const int wave, const int lane) {
  constexpr int CH = kPsChunk;
  constexpr int SGB = tile_stride(32 * NCS);          // row stride of the stationary tile (bytes)
  constexpr int PPS = SGB / 16;                       // 16-byte pieces per (padded) stationary row
  constexpr int NST = 16 * SGB / 1024;                // DMA instructions for this wave's 16 stationary rows
  constexpr int NIX = NW + 1;                         // index DMA instructions per chunk (NW offsets + output rows)
  constexpr int IDXB = NW * 512 + 64;                 // bytes per index buffer
  const View &v = a.v;
  const int K = v.K;
  char *Stile = smem;                                 // [2][CH][SGB]
  char *Aring = smem + wave_off;                      // [3][NW][1 KB] gathered tiles
  char *idxb = Aring + 3 * NW * 1024;                 // [2][IDXB]: [NW][128] neighbour rows, [16] stationary rows
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address of smem

  const int c_begin = wlane * a.chunks_per_lane;
  const int c_end = min(c_begin + a.chunks_per_lane, a.n_chunks);

  f32x16 acc[NW][NCS];
#pragma unroll
  for (int o = 0; o < NW; ++o)
#pragma unroll
    for (int b = 0; b < NCS; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[o][b][r] = 0.f;

  constexpr unsigned kOOB = 0xfffff000u;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.G), 0, (int)a.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.S), 0, (int)a.s_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(v.nbr), 0, (int)a.nbr_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(v.out_row ? v.out_row : v.nbr), 0,
                                                                        (int)(v.out_row ? a.orow_bytes : 0u), 0x00020000);
  const bool has_orow = v.out_row != nullptr;
  const unsigned g_row_b = (unsigned)a.g_ld * 2u, s_row_b = (unsigned)a.s_ld * 2u;
  // gather: lane -> (row lane/4 of the 16-position group, 16-byte piece lane%4 of the 64-byte channel slice)
  const int g_row = lane >> 2;
  const unsigned g_ch = (cg0 + (lane & 3) * 8 + 8 <= a.cg_real) ? (unsigned)(cg0 + (lane & 3) * 8) * 2u : kOOB;
  // tr read: 16-lane group g = lane>>4: cb = g&1 (16-channel half), h = g>>1 (positions 8h..8h+7); lane i = lane&15
  // supplies the 8-byte address (row 8h + i/4 (+4 for the second read), channel 32*blk + 16*cb + 4*(i%4))
  const int g16 = lane >> 4, i16 = lane & 15;
  const int tr_row = 8 * (g16 >> 1) + (i16 >> 2);
  const int tr_col = 16 * (g16 & 1) + 4 * (i16 & 3);
  const unsigned a_tr = lds0 + (unsigned)wave_off + (unsigned)(tr_row * 64 + tr_col * 2);          // + slot * NW KB + o KB
  const unsigned s_tr = lds0 + (unsigned)(tr_row * SGB + tr_col * 2);                               // + buf * CH*SGB + 16 s SGB
  const unsigned idx_rd = lds0 + (unsigned)wave_off + (unsigned)(3 * NW * 1024);

  uint32_t actbits = 0;     // bit 4*slot + o: gathered tile (slot, o) has at least one neighbour
  int islot = 0;            // ring slot the next issued gather group goes to (3 slots)

  // ---- issue side
  auto issue_index = [&](int c2) __attribute__((always_inline)) {   // kernel-map rows + output rows of chunk c2
    char *dst = idxb + (c2 & 1) * IDXB;
    const bool in = c2 < c_end;
#pragma unroll
    for (int o = 0; o < NW; ++o) {
      const unsigned off = in ? (unsigned)(((int64_t)koff[o] * v.n_pad + (int64_t)c2 * CH + (lane & 31) * 4) * 4) : kOOB;
      if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_n, LGS_AS3(dst + o * 512), 16, off, 0, 0, 0);
    }
    const unsigned off = in ? (unsigned)(((int64_t)c2 * CH + 16 * wave + (lane & 3) * 4) * 4) : kOOB;
    if (lane < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, LGS_AS3(dst + NW * 512), 16, off, 0, 0, 0);
  };
  auto issue_stage = [&](int c1) __attribute__((always_inline)) {   // this wave's 16 rows of the stationary tile of chunk c1
    char *dst = Stile + (c1 & 1) * (CH * SGB) + 16 * wave * SGB;
    const unsigned srow_rd = idx_rd + (unsigned)((c1 & 1) * IDXB + NW * 512);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int q = j * 64 + lane, row = q / PPS, cp = q % PPS;
      int32_t sr;
      if (has_orow) {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(sr) : "v"(srow_rd + (unsigned)row * 4u) : "memory");
      } else {
        const int64_t p = (int64_t)c1 * CH + 16 * wave + row;
        sr = p < v.n_out ? (int32_t)p : -1;
      }
      const bool ok = c1 < c_end && sr >= 0 && cp < 4 * NCS && cs0 + cp * 8 + 8 <= a.cs_real;
      const unsigned off = ok ? (unsigned)sr * s_row_b + (unsigned)(cs0 + cp * 8) * 2u : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, LGS_AS3(dst + j * 1024), 16, off, 0, 0, 0);
    }
  };
  // One step of the stream: the kernel-map rows of the NEXT gather group (nc, ns) are requested from LDS, then -- once
  // the DMA of group (cc, s) has landed (vmcnt) -- its fragments; ONE lgkmcnt wait covers both, the next group's DMA
  // is issued and only then the MFMAs of group (cc, s) run: the LDS round trip is paid once per step, and the new
  // gathers are in flight during the MFMAs.  nvm = DMA instructions that may still be outstanding behind group (cc, s).
  int cslot = 0;
  auto step = [&](int cc, int s, int nc, int ns, auto nvm_tag) __attribute__((always_inline)) {
    constexpr int NVM = decltype(nvm_tag)::value;
    // four-offset waves (192 accumulators) have no registers to hold the next group's rows across the fragment reads:
    // they read them AFTER their MFMAs (one more LDS round trip per step)
    constexpr bool FUSE = NW <= 3;
    // (1) kernel-map rows of the next group
    const unsigned rd = idx_rd + (unsigned)((nc & 1) * IDXB + (16 * ns + g_row) * 4);
    int32_t r[NW];
    if constexpr (FUSE) {
#pragma unroll
      for (int o = 0; o < NW; ++o) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[o]) : "v"(rd), "n"(o * 512) : "memory");
    }
    // (2) fragments of this group
    LGS_VMCNT(NVM);
    const uint32_t m = (actbits >> (4 * cslot)) & 0xfu;
    u32x2 fbr[NCS][2], far_[NW][2];
    constexpr int NA0 = NW <= 3 ? NW : 1;              // A tiles read with the B fragments (all of them when registers allow)
    if (m != 0) {                                      // wave-uniform
      const unsigned sb = s_tr + (unsigned)((cc & 1) * (CH * SGB) + 16 * s * SGB);
      const unsigned ab = a_tr + (unsigned)(cslot * (NW * 1024));
#pragma unroll
      for (int b = 0; b < NCS; ++b) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(fbr[b][0]) : "v"(sb), "n"(64 * b) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(fbr[b][1]) : "v"(sb), "n"(64 * b + 4 * SGB) : "memory");
      }
#pragma unroll
      for (int o = 0; o < NA0; ++o) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(far_[o][0]) : "v"(ab), "n"(o * 1024) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(far_[o][1]) : "v"(ab), "n"(o * 1024 + 256) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (FUSE) {
#pragma unroll
      for (int o = 0; o < NW; ++o) asm volatile("" : "+v"(r[o]));
    }
    if (m != 0) {
#pragma unroll
      for (int b = 0; b < NCS; ++b) { asm volatile("" : "+v"(fbr[b][0])); asm volatile("" : "+v"(fbr[b][1])); }
#pragma unroll
      for (int o = 0; o < NA0; ++o) { asm volatile("" : "+v"(far_[o][0])); asm volatile("" : "+v"(far_[o][1])); }
    }
    // (3) DMA of the next group
    auto issue_next = [&]() __attribute__((always_inline)) {
      uint32_t mm = 0;
      char *dst = Aring + islot * (NW * 1024);
#pragma unroll
      for (int o = 0; o < NW; ++o) {
        // a tile without any neighbour is still "fetched" (every lane out of range: zeros, no memory traffic): the DMA
        // stream stays static and every wait is a compile-time vmcnt.  Skipping such tiles with a run-time vmcnt (scalar
        // switch over the count) was built and measured SLOWER (L0 96->96: 0.67 vs 0.63 ms).
        const bool ok = nc < c_end && r[o] >= 0;
        if (__ballot(ok) != 0ull) mm |= 1u << o;
        const unsigned off = (ok && g_ch != kOOB) ? (unsigned)r[o] * g_row_b + g_ch : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, LGS_AS3(dst + o * 1024), 16, off, 0, 0, 0);
      }
      actbits = (actbits & ~(0xfu << (4 * islot))) | (mm << (4 * islot));
      islot = islot == 2 ? 0 : islot + 1;
    };
    if constexpr (FUSE) issue_next();
    // (4) MFMAs of this group
    if (m != 0) {
      const unsigned ab = a_tr + (unsigned)(cslot * (NW * 1024));
      bf16x8 fb[NCS];
#pragma unroll
      for (int b = 0; b < NCS; ++b) {
        u32x4 pk; pk.x = fbr[b][0].x; pk.y = fbr[b][0].y; pk.z = fbr[b][1].x; pk.w = fbr[b][1].y;
        fb[b] = __builtin_bit_cast(bf16x8, pk);
      }
#pragma unroll
      for (int o = 0; o < NW; ++o) {
        // four-offset waves: tile o+1 is requested in front of the MFMAs of tile o (two A fragments live at most)
        if constexpr (NA0 < NW) {
          if (o + 1 >= NA0 && o + 1 < NW) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(far_[o + 1 < NW ? o + 1 : 0][0]) : "v"(ab), "n"((o + 1) * 1024) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(far_[o + 1 < NW ? o + 1 : 0][1]) : "v"(ab), "n"((o + 1) * 1024 + 256) : "memory");
          }
        }
        if ((m >> o) & 1u) {                            // wave-uniform
          u32x4 pk; pk.x = far_[o][0].x; pk.y = far_[o][0].y; pk.z = far_[o][1].x; pk.w = far_[o][1].y;
          const bf16x8 fa = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
          for (int b = 0; b < NCS; ++b) acc[o][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[b], acc[o][b], 0, 0, 0);
        }
        if constexpr (NA0 < NW) {
          if (o + 1 >= NA0 && o + 1 < NW) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(far_[o + 1 < NW ? o + 1 : 0][0])); asm volatile("" : "+v"(far_[o + 1 < NW ? o + 1 : 0][1]));
          }
        }
      }
    }
    if constexpr (!FUSE) {
#pragma unroll
      for (int o = 0; o < NW; ++o) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[o]) : "v"(rd), "n"(o * 512) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int o = 0; o < NW; ++o) asm volatile("" : "+v"(r[o]));
      issue_next();
    }
    cslot = cslot == 2 ? 0 : cslot + 1;
  };
  // prologue form: issue a gather group without consuming anything
  auto issue_gather = [&](int cc, int s) __attribute__((always_inline)) {
    const unsigned rd = idx_rd + (unsigned)((cc & 1) * IDXB + (16 * s + g_row) * 4);
    int32_t r[NW];
#pragma unroll
    for (int o = 0; o < NW; ++o) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[o]) : "v"(rd), "n"(o * 512) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int o = 0; o < NW; ++o) asm volatile("" : "+v"(r[o]));
    uint32_t m = 0;
    char *dst = Aring + islot * (NW * 1024);
#pragma unroll
    for (int o = 0; o < NW; ++o) {
      const bool ok = cc < c_end && r[o] >= 0;
      if (__ballot(ok) != 0ull) m |= 1u << o;
      const unsigned off = (ok && g_ch != kOOB) ? (unsigned)r[o] * g_row_b + g_ch : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, LGS_AS3(dst + o * 1024), 16, off, 0, 0, 0);
    }
    actbits = (actbits & ~(0xfu << (4 * islot))) | (m << (4 * islot));
    islot = islot == 2 ? 0 : islot + 1;
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ---- prologue: indices of the first two chunks, then the first stationary tile and two gather groups
  issue_index(c_begin);
  issue_index(c_begin + 1);
  LGS_VMCNT(0);
  issue_stage(c_begin);
  issue_gather(c_begin, 0);
  issue_gather(c_begin, 1);
  LGS_VMCNT(2 * NW);           // the stationary rows have landed (the two gather groups may still be in flight)
  barrier();

  for (int cc = c_begin; cc < c_end; ++cc) {
    // program order of the DMA stream:  ... G(cc,0) G(cc,1) | ST(cc+1) G(cc,2) .. G(cc,7) IX(cc+2) G(cc+1,0) G(cc+1,1) | ...
    // issue_stage(cc+1) reads the output rows of chunk cc+1 that IX(cc+1) fetched: IX(cc+1) was issued in the previous
    // iteration with only G(cc,0), G(cc,1) behind it, so "at most 2 NW outstanding" means it has landed
    LGS_VMCNT(2 * NW);
    issue_stage(cc + 1);
    // step(cc, s, next group, N): N = DMA instructions behind group (cc, s) at its wait (the next group is issued after it)
    step(cc, 0, cc, 2, std::integral_constant<int, NW + NST>());      // behind G0: G1, ST
    step(cc, 1, cc, 3, std::integral_constant<int, NST + NW>());      // behind G1: ST, G2   (ST lands before G2: covered)
    step(cc, 2, cc, 4, std::integral_constant<int, NW>());            // behind Gs: G(s+1)
    step(cc, 3, cc, 5, std::integral_constant<int, NW>());
    step(cc, 4, cc, 6, std::integral_constant<int, NW>());
    step(cc, 5, cc, 7, std::integral_constant<int, NW>());
    issue_index(cc + 2);       // buffer cc % 2: its last reader was the step that issued G(cc,7)
    step(cc, 6, cc + 1, 0, std::integral_constant<int, NW + NIX>());  // behind G6: G7, IX
    step(cc, 7, cc + 1, 1, std::integral_constant<int, NIX + NW>());  // behind G7: IX, G(cc+1,0)
    barrier();                 // stationary tile of chunk cc+1 complete (landed before group 2's wait); this one is free
  }
  LGS_VMCNT(0);

  // ---- partial[lane][k][cg][cs]: D[i = cg channel][j = cs channel], lane holds column j = lane&31, rows (r&3)+8(r>>2)+4h
  const int vx = lane & 31, h = lane >> 5;
#pragma unroll
  for (int o = 0; o < NW; ++o) {
    const int k = koff[o];
    float *dst = a.partial + (((int64_t)wlane * K + k) * a.cg_pad) * (int64_t)a.cs_pad;
#pragma unroll
    for (int b = 0; b < NCS; ++b) {
      const int cj = cs0 + 32 * b + vx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cg0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (ci < a.cg_pad && cj < a.cs_pad) dst[(int64_t)ci * a.cs_pad + cj] = acc[o][b][r];
      }
    }
  }
}

// KIND 27: 3^3 map, waves own 3/3/3/3/4/4/4/3 offsets;  KIND 8: 2^3 map, one offset per wave
template <int KIND, int NCS>
__global__ __launch_bounds__(512, 2) void k_wgrad_ps(PsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as a SCALAR: every LDS destination of the DMA stream (M0) and every ring address derives from it, and
  // hipcc otherwise re-derives them from the vector thread id with a v_readfirstlane per DMA instruction
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned bx = blockIdx.x, xcd = bx & 7u, tq = bx >> 3;
  const int n_sl = a.n_cg * a.n_cs;
  // xcd_map: the slices of one lane share an XCD (= its L2); otherwise (more slices than an XCD has CUs) plain order
  const int slice = (int)((a.xcd_map ? tq : bx) % (unsigned)n_sl);
  const int wlane = a.xcd_map ? (int)(tq / (unsigned)n_sl) * 8 + (int)xcd : (int)(bx / (unsigned)n_sl);
  if (wlane >= a.n_lanes) return;
  const int cg0 = (slice % a.n_cg) * 32, cs0 = (slice / a.n_cg) * 32 * NCS;
  constexpr int SGB = tile_stride(32 * NCS);
  constexpr int S_BYTES = 2 * kPsChunk * SGB;
  if constexpr (KIND == 27) {
    // waves 0-3 and 7 own three offsets, waves 4-6 four: their LDS regions are sized accordingly
    const int n4 = wave <= 4 ? 0 : (wave <= 7 ? wave - 4 : 3);          // four-offset waves in front of this one
    const int wave_off = S_BYTES + (wave - n4) * ps_wave_lds(3) + n4 * ps_wave_lds(4);
    if (wave >= 4 && wave <= 6) {
      const int koff[4] = {kPsOff27[wave][0], kPsOff27[wave][1], kPsOff27[wave][2], kPsOff27[wave][3]};
      ps_wave<4, NCS>(a, smem, wave_off, koff, wlane, cg0, cs0, wave, lane);
    } else {
      const int koff[3] = {kPsOff27[wave][0], kPsOff27[wave][1], kPsOff27[wave][2]};
      ps_wave<3, NCS>(a, smem, wave_off, koff, wlane, cg0, cs0, wave, lane);
    }
  } else {
    const int koff[1] = {wave};
    ps_wave<1, NCS>(a, smem, S_BYTES + wave * ps_wave_lds(1), koff, wlane, cg0, cs0, wave, lane);
  }
}

// gw[k][ci][co] = sum over the lane slabs, fixed order.  Slabs hold D_k[cg][cs]; transpose = 1 when the stationary
// operand was the op's INPUT (transposed conv: D_k = gw[k]^T).  One thread = 4 consecutive cs channels.
__global__ __launch_bounds__(256) void k_wgrad_reduce_ps(const float *__restrict__ partial, int S, int K, int cg_pad, int cs_pad,
                                                         int cg, int cs, int transpose, float *__restrict__ gw) {
  const int cq = (cs + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * cg * cq;
  if (idx >= total) return;
  const int q = (int)(idx % cq);
  const int ig = (int)((idx / cq) % cg);
  const int k = (int)(idx / ((int64_t)cq * cg));
  const int64_t slab = (int64_t)K * cg_pad * cs_pad;
  const float *src = partial + ((int64_t)k * cg_pad + ig) * cs_pad + 4 * q;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int x = 0;
  for (; x + 4 <= S; x += 4) {
    const float4 v0 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 0) * slab);
    const float4 v1 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 1) * slab);
    const float4 v2 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 2) * slab);
    const float4 v3 = *reinterpret_cast<const float4 *>(src + (int64_t)(x + 3) * slab);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; x < S; ++x) {
    const float4 v0 = *reinterpret_cast<const float4 *>(src + (int64_t)x * slab);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
  }
  const float r[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                      (a0.w + a1.w) + (a2.w + a3.w)};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int is = 4 * q + j;
    if (is >= cs) continue;
    // plain: gw[k][ci = ig][co = is];  transposed: gw[k][ci = is][co = ig]
    if (!transpose) gw[((int64_t)k * cg + ig) * cs + is] = r[j];
    else gw[((int64_t)k * cs + is) * cg + ig] = r[j];
  }
}

// ------------------------------------------------------------------------------------ host side
struct WgradPlan {
  int S;          // partial slots
  int64_t span;   // positions per slot
  int cin_pad, cout_pad;
  int ncb;        // fp32 path: co blocks per wave
  int nci, nco;   // bf16 path: wave tile in 32-channel blocks
  int n_ci_tasks, n_co_tasks;
  int kpw;        // bf16 path: kernel offsets per workgroup (one per wave)
  int n_ranges;   // bf16 path: position ranges of `span` voxels, dealt round-robin to the S lanes
};

inline WgradPlan wgrad_plan(const View &v, int cin, int cout, int dtype) {
  WgradPlan p{};
  p.cin_pad = pad32(cin);
  p.cout_pad = pad32(cout);
  const int nbi = p.cin_pad / 32, nbo = p.cout_pad / 32;
  int64_t per = (int64_t)v.K * p.cin_pad * p.cout_pad * 4;
  if (dtype == LGS_BF16) {
    // wave tile <= 4 x 3 blocks (192 accumulator registers)
    p.nci = nbi >= 4 ? 4 : nbi;
    p.nco = nbo >= 3 ? 3 : nbo;
    if (nbo % 3 != 0 && nbo % 2 == 0 && nbo >= 2) p.nco = 2;
    if (nbo % 4 == 0 && p.nci <= 3) p.nco = 4;
    // tiles whose accumulators + a 2-deep gather ring exceed ~192 registers would run one wave per SIMD: halve them
    if (16 * p.nci * p.nco + 8 * (p.nci + p.nco) > 192) { if (p.nci == 4) p.nci = 2; else p.nco = 2; }
    p.n_ci_tasks = (nbi + p.nci - 1) / p.nci;
    p.n_co_tasks = (nbo + p.nco - 1) / p.nco;
    // large maps: one offset per wave (4 offsets share a range's rows in L1/L2); small maps and the grouped views
    // (whose positions are sorted by offset, so a range holds a single offset): one offset per workgroup, the four
    // waves split the range
    // (kpw is 4 or 1 only: the accumulator fold of sub-range waves below shares ONE LDS tile per workgroup)
    p.kpw = (v.n_pad >= 65536 && v.KS > 1 && v.K >= 4) ? 4 : 1;
    // smaller maps: ranges of 1024 positions, dealt round-robin to S "lanes" of co-resident workgroups (see the
    // kernel): S is what fits the chip's workgroup slots (LDS bound, <= 4 per CU), a multiple of 8 (XCD pinning), and
    // every lane costs one partial slab of K x Cin x Cout floats
    const int64_t KG = (v.K + p.kpw - 1) / p.kpw, tasks = (int64_t)p.n_ci_tasks * p.n_co_tasks;
    const int wave_bytes = 2 * kQ * 4 + 16 * tile_stride(32 * p.nci) + 16 * tile_stride(32 * p.nco);
    int wg_per_cu = (160 * 1024) / (4 * wave_bytes);
    if (wg_per_cu > 4) wg_per_cu = 4;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const int64_t slots = 256 * (int64_t)wg_per_cu;
    int64_t range = 4096, n_ranges = (v.n_pad + range - 1) / range, lanes = n_ranges;
    if (n_ranges * KG * tasks >= 2 * slots && n_ranges * per <= (1ll << 30)) {
      // big maps: one 4096-position range per workgroup, >= 2 rounds of workgroups over the chip.  The offsets of a
      // 3^3 map differ 4x in density (centre 100 %, corners ~25 %), so dynamic dispatch is what balances the waves.
    } else {
      range = 1024;
      n_ranges = (v.n_pad + range - 1) / range;
      lanes = slots / (KG * tasks);
      lanes = lanes / 8 * 8;
      if (lanes < 8) lanes = 8;
      while (lanes > 8 && lanes * per > (1ll << 30)) lanes -= 8;
      if (n_ranges <= lanes + lanes / 4) {
        lanes = n_ranges;                       // mild over-subscription beats a half-empty second round
      } else {                                  // equalise the rounds per lane
        const int64_t rounds = (n_ranges + lanes - 1) / lanes;
        const int64_t l2 = ((n_ranges + rounds - 1) / rounds + 7) / 8 * 8;
        if (l2 < lanes) lanes = l2;
      }
    }
    p.span = range;
    p.n_ranges = (int)n_ranges;
    p.S = (int)(lanes > 0 ? lanes : 1);
  } else {
    p.ncb = (nbo % 4 == 0) ? 4 : (nbo % 3 == 0) ? 3 : (nbo % 2 == 0) ? 2 : 1;
    if (p.ncb == 1 && nbo >= 5) {
      // 5 or 7 output blocks (the 200-class head: 7): one block per workgroup would stage the input rows seven times over; two
      // four-block column tiles over a gradient image padded to 8 blocks stage them twice (the reduction drops the padding)
      p.ncb = 4;
      p.cout_pad = (nbo + 3) / 4 * 128;
      per = (int64_t)v.K * p.cin_pad * p.cout_pad * 4;
    }
    int64_t chunks = (v.n_pad + kWgChunk - 1) / kWgChunk;
    int64_t S = chunks / 4;
    if (S < 1) S = 1;
    if (S > 128) S = 128;     // (64 until round 5: 27 x 64 workgroups of very unequal work -- the centre offset has ~7 x a corner's pairs -- left a long tail)
    while (S > 1 && S * per > (1ll << 30)) S /= 2;
    int64_t cps = (chunks + S - 1) / S;
    p.span = cps * kWgChunk;
    p.S = (int)((v.n_pad + p.span - 1) / p.span);
    if (p.S < 1) p.S = 1;
  }
  return p;
}

// ---- plan of the position-stationary kernel
struct PsPlan {
  bool ok = false;
  int noff = 0, ncs = 0;
  int cg_pad = 0, cs_pad = 0, n_cg = 0, n_cs = 0, n_lanes = 0, cpl = 0, n_chunks = 0, xcd_map = 1;
  int64_t partial_bytes = 0;
};
inline PsPlan ps_plan(const View &v, int cg, int cs, int cus_override = 0) {
  PsPlan p;
  if (!(v.K == 27 || v.K == 8) || v.KS != v.K || v.nbr == nullptr || v.n_pad < kPsChunk || v.n_pad % kPsChunk != 0) return p;
  if (cg % 8 != 0 || cs % 8 != 0) return p;
  p.cg_pad = pad32(cg); p.cs_pad = pad32(cs);
  const int nbs = p.cs_pad / 32;
  p.noff = v.K == 27 ? 4 : 1;
  // K = 27: <= 3 blocks (4 offsets x 3 blocks = 192 accumulators).  Every stationary slice re-gathers all neighbour rows, and
  // gather instructions are what a wide-channel launch is made of (PMC at 512 x 512: 83 M vector-memory instructions
  // for 68 M MFMAs): from 8 blocks on, 3-block slices are used even when the last one is partly padding (16 blocks:
  // 6 slices instead of 8, 12 % padded MFMAs)
  const bool wide3 = tune(T_PS_WIDE3) != 0;
  // (512 -> 512 at L0: 18.7 -> 15.0 ms, 640 -> 512: 33.3 -> 19.3 ms; at 8 blocks only when the slices exceed an XCD's CUs
  // anyway -- 256 x 256 keeps its 32 XCD-local two-block slices: 1.21 vs 1.34 ms at L1)
  const bool three = nbs % 3 == 0 || (wide3 && (nbs >= 12 || (nbs >= 8 && (p.cg_pad / 32) * ((nbs + 2) / 3) > 32)));
  if (v.K == 27) p.ncs = nbs == 1 ? 1 : (three ? 3 : (nbs % 2 == 0 ? 2 : 3));
  else p.ncs = nbs <= 4 ? nbs : (nbs % 4 == 0 ? 4 : (nbs % 3 == 0 ? 3 : 4));
  p.n_cg = p.cg_pad / 32;
  p.n_cs = (nbs + p.ncs - 1) / p.ncs;
  p.n_chunks = (int)(v.n_pad / kPsChunk);
  // One (K = 27: LDS / register bound) or two workgroups per CU.  Workgroup b runs on XCD b % 8 and the slices of a lane
  // are placed on one XCD (they share its rows in that L2), so the lanes of an XCD must fit ITS 32 CUs: with 3 slices
  // 85 lanes would put 33 workgroups on five of the XCDs and the 33rd runs alone in a second round (measured 2x).
  const int lds_wg = 2 * kPsChunk * tile_stride(32 * p.ncs) + (v.K == 27 ? 5 * ps_wave_lds(3) + 3 * ps_wave_lds(4) : 8 * ps_wave_lds(1));
  const int wg_per_cu = v.K == 27 ? 1 : (2 * lds_wg <= 160 * 1024 ? 2 : 1);
  // CUs per XCD the kernel fills: 20 of 32 (tuning knob PS_CUS).  A workgroup owns its CU (all registers, all LDS) for the
  // whole launch, and the kernel runs next to the dgrad / BatchNorm chain of the compute stream: with every CU taken, each
  // compute-stream kernel waits for weight-gradient workgroups to retire before it gets anywhere.  Leaving 12 CUs per XCD makes
  // the weight gradients ~1.3 x longer on their own stream (which has the slack) and the 8-scene step 0.7 ms shorter
  // (32: 29.97, 28: 29.6, 24: 29.4, 20: 29.27, 16: 29.36 ms; `finalize`, the side stream's tail, stays 0.40 ms down to 20)
  const int cus_env = cus_override > 0 ? cus_override : (int)tune(T_PS_CUS);
  const int per_xcd = (cus_env >= 4 && cus_env <= 32 ? cus_env : 32) * wg_per_cu, n_sl = p.n_cg * p.n_cs;
  p.xcd_map = n_sl <= per_xcd ? 1 : 0;
  int lanes = p.xcd_map ? 8 * (per_xcd / n_sl) : (8 * per_xcd) / n_sl;
  if (lanes < 1) lanes = 1;
  // more slices than an XCD has CUs: workgroups go out in plain order, so make their number a whole multiple of the
  // chip (96 slices x 2 lanes would leave a quarter of the CUs idle; x 8 lanes = three full rounds)
  if (!p.xcd_map && wide3)
    while ((lanes * n_sl) % (8 * per_xcd) != 0 && lanes < 16) ++lanes;
  if (lanes > p.n_chunks) lanes = p.n_chunks;
  p.cpl = (p.n_chunks + lanes - 1) / lanes;
  p.n_lanes = (p.n_chunks + p.cpl - 1) / p.cpl;         // every lane owns at least one chunk
  p.partial_bytes = (int64_t)p.n_lanes * v.K * p.cg_pad * p.cs_pad * 4;
  p.ok = p.partial_bytes <= (2ll << 30);
  return p;
}
inline bool ps_enabled() {
  return tune(T_WGRAD_PS) != 0;   // debugging knob: 0 forces the pair-list kernel
}

// bytes of the zero-padded input copy the fp32 path makes for input widths off the 4-channel grid (the 3-channel colour input)
inline int64_t f32_pad_bytes(const View &v, int cin, int cout) {
  return (cin % 4 != 0 && cout % 4 == 0 && tune(T_WGRAD_F32_LDS) != 0) ? align256(v.n_in * (int64_t)((cin + 3) / 4 * 4) * 4) : 0;
}

int64_t wgrad_workspace_bytes(const lgs_kmap *km, int cin, int cout, int dtype) {
  WgradPlan a = wgrad_plan(km->fwd, cin, cout, dtype), b = wgrad_plan(km->bwd, cin, cout, dtype);
  int64_t per = (int64_t)km->K * pad32(cin) * (a.cout_pad > b.cout_pad ? a.cout_pad : b.cout_pad) * 4;
  int64_t bytes = align256((int64_t)(a.S > b.S ? a.S : b.S) * per) + 256;
  if (dtype == LGS_BF16) {   // position-stationary kernel: forward direction (gathered = in) and transposed (gathered = gout)
    PsPlan f = ps_plan(km->fwd, cin, cout), t = ps_plan(km->fwd, cout, cin);
    const int64_t pb = (f.ok ? f.partial_bytes : 0) > (t.ok ? t.partial_bytes : 0) ? (f.ok ? f.partial_bytes : 0) : (t.ok ? t.partial_bytes : 0);
    if (align256(pb) + 256 > bytes) bytes = align256(pb) + 256;
    const int64_t wb = wgrad_wide_workspace_bytes(km->fwd, cin, cout);
    if (wb > bytes) bytes = wb;
  }
  if (dtype == LGS_F32) {   // zero-padded input copy behind the partial slabs (conv_wgrad_f32path)
    const int64_t pf = f32_pad_bytes(km->fwd, cin, cout), pb = f32_pad_bytes(km->bwd, cin, cout);
    bytes += (pf > pb ? pf : pb) + 256;
  }
  if (dtype == LGS_BF16 && cin % 8 != 0) {
    int64_t nmax = km->fwd.n_in > km->bwd.n_in ? km->fwd.n_in : km->bwd.n_in;
    const int c8 = (cin + 7) / 8 * 8;
    const int64_t padded = align256(nmax * (int64_t)c8 * 2);
    bytes += padded;
    // ... or the position-stationary kernel on the padded rows (lgs_conv_wgrad): partial slabs | padded input
    const PsPlan f8 = ps_plan(km->fwd, c8, cout, 32);
    if (f8.ok && align256(f8.partial_bytes) + 256 + padded + 256 > bytes) bytes = align256(f8.partial_bytes) + 256 + padded + 256;
  }
  if (dtype == LGS_BF16 && cout % 8 != 0) {
    int64_t nmax = km->fwd.n_out > km->bwd.n_out ? km->fwd.n_out : km->bwd.n_out;
    bytes += align256(nmax * (int64_t)((cout + 7) / 8 * 8) * 2);
  }
  return bytes;
}

template <int NCI, int NCO>
int launch_wgrad_bf16(const View &v, const WgradPlan &p, const bf16_t *in, int cin, const bf16_t *go, int cout,
                      float *partial, hipStream_t s) {
  constexpr int CA = 32 * NCI, CG = 32 * NCO;
  const uint64_t in_b = (uint64_t)v.n_in * cin * 2, go_b = (uint64_t)v.n_out * cout * 2;
  LGS_REQUIRE(in_b < 0xfffff000ull && go_b < 0xfffff000ull,
              "bf16 wgrad: a feature tensor of 4 GiB or more is beyond the 32-bit buffer-descriptor path");
  // two waves per SIMD (<= 256 registers) whenever accumulators + a >= 2-deep gather ring fit: the LDS/MFMA part of a
  // group is a serial chain of latencies that only a second wave can hide (1 wave/SIMD ran 1.4x slower at 96x96)
  constexpr int ACC = 16 * NCI * NCO, RING1 = 4 * (NCI + NCO);
  constexpr int OCC = (ACC + 2 * RING1 <= 192) ? 2 : 1;
  constexpr int D = OCC == 1 ? 3 : ((ACC + 4 * RING1 <= 192) ? 4 : (ACC + 3 * RING1 <= 192) ? 3 : 2);
  constexpr int WAVE_BYTES = 2 * kQ * 4 + 16 * tile_stride(CA) + 16 * tile_stride(CG);
  static_assert(4 * WAVE_BYTES >= NCI * NCO * 16 * 64 * 4, "staging LDS must hold one accumulator tile");
  const int n_tasks = p.n_ci_tasks * p.n_co_tasks;
  const int n_ranges = p.n_ranges, n_lanes = p.S;
  const int KG = (v.K + p.kpw - 1) / p.kpw;
  const unsigned nblocks = (unsigned)(((n_lanes + 7) / 8) * 8 * KG * n_tasks);
  LGS_KLAUNCH((k_wgrad_bf16<NCI, NCO, D, OCC>), dim3(nblocks), 256, 4 * WAVE_BYTES, s, v, in, cin, go, cout, p.cin_pad,
                     p.cout_pad, p.span, p.kpw, p.n_ci_tasks, n_tasks, n_ranges, n_lanes, partial, (unsigned)in_b, (unsigned)go_b);
  return 0;
}

template <int KIND, int NCS>
int launch_wgrad_ps(const PsArgs &a, const PsPlan &p, hipStream_t s) {
  constexpr int SGB = tile_stride(32 * NCS);
  constexpr int LDS = 2 * kPsChunk * SGB + (KIND == 27 ? 5 * ps_wave_lds(3) + 3 * ps_wave_lds(4) : 8 * ps_wave_lds(1));
  static_assert(LDS <= 160 * 1024, "k_wgrad_ps: LDS budget of one CU");
  static bool attr_set = false;
  if (!attr_set) {
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_ps<KIND, NCS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const unsigned nblocks = (unsigned)((p.xcd_map ? ((p.n_lanes + 7) / 8) * 8 : p.n_lanes) * p.n_cg * p.n_cs);
  LGS_KLAUNCH((k_wgrad_ps<KIND, NCS>), dim3(nblocks), 512, LDS, s, a);
  return 0;
}

// gw[K][cin][cout] through the position-stationary kernel.  v = the map's forward view (3^3, or the coarse-stationary
// 2^3 view); transposed = 0: gathered operand = in (rows of v's input side), stationary = gout;  transposed = 1 (the
// transposed conv that reuses the strided conv's map): gathered = gout (fine rows), stationary = in (coarse rows).
int conv_wgrad_ps(const View &v, int transposed, const bf16_t *in, int cin, const bf16_t *go, int cout, float *gw, void *workspace,
                  hipStream_t s, bool *done, int in_ld, int cin_out = -1, int cus_override = 0) {
  // cin_out (> 0, forward direction only): `in` was zero-padded to cin channels by the caller; gw has cin_out input channels
  *done = false;
  if (!ps_enabled()) return 0;
  const int cg = transposed ? cout : cin, cs = transposed ? cin : cout;
  const PsPlan p = ps_plan(v, cg, cs, cus_override);
  if (!p.ok) return 0;
  const int64_t g_rows = v.n_in, s_rows = v.n_out;
  const int g_ld = transposed ? cg : (in_ld > 0 ? in_ld : cg), s_ld = transposed ? (in_ld > 0 ? in_ld : cs) : cs;
  if ((g_ld * 2) % 16 != 0 || (s_ld * 2) % 16 != 0) return 0;
  const uint64_t g_b = (uint64_t)g_rows * g_ld * 2, s_b = (uint64_t)s_rows * s_ld * 2, n_b = (uint64_t)v.KS * v.n_pad * 4, o_b = (uint64_t)v.n_pad * 4;
  if (!(g_b < 0xfffff000ull && s_b < 0xfffff000ull && n_b < 0xfffff000ull)) return 0;   // beyond the 32-bit descriptor path
  PsArgs a;
  a.v = v; a.v.mirror = 0;
  a.G = transposed ? go : in; a.S = transposed ? in : go;
  a.g_ld = g_ld; a.s_ld = s_ld;
  a.cg_real = cg; a.cs_real = cs; a.cg_pad = p.cg_pad; a.cs_pad = p.cs_pad; a.n_cg = p.n_cg; a.n_cs = p.n_cs;
  a.n_lanes = p.n_lanes; a.chunks_per_lane = p.cpl; a.n_chunks = p.n_chunks; a.xcd_map = p.xcd_map;
  a.partial = reinterpret_cast<float *>(workspace);
  a.g_bytes = (unsigned)g_b; a.s_bytes = (unsigned)s_b; a.nbr_bytes = (unsigned)n_b; a.orow_bytes = (unsigned)o_b;
  int rc = 0;
  if (p.noff == 4) {
    if (p.ncs == 1) rc = launch_wgrad_ps<27, 1>(a, p, s);
    else if (p.ncs == 2) rc = launch_wgrad_ps<27, 2>(a, p, s);
    else rc = launch_wgrad_ps<27, 3>(a, p, s);
  } else {
    if (p.ncs == 1) rc = launch_wgrad_ps<8, 1>(a, p, s);
    else if (p.ncs == 2) rc = launch_wgrad_ps<8, 2>(a, p, s);
    else if (p.ncs == 3) rc = launch_wgrad_ps<8, 3>(a, p, s);
    else rc = launch_wgrad_ps<8, 4>(a, p, s);
  }
  if (rc) return rc;
  const int cg_o = (!transposed && cin_out > 0) ? cin_out : cg;
  const int64_t total = (int64_t)v.K * cg_o * ((cs + 3) / 4);
  LGS_KLAUNCH(k_wgrad_reduce_ps, (unsigned)((total + 255) / 256), 256, 0, s, a.partial, p.n_lanes, v.K, p.cg_pad, p.cs_pad, cg_o, cs,
                     transposed, gw);
  LGS_HIP(hipGetLastError());
  *done = true;
  return 0;
}

__global__ void k_pad_rows_bf16(const bf16_t *__restrict__ src, int64_t n, int c, int cpad, bf16_t *__restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * cpad) return;
  int64_t r = i / cpad;
  int ch = (int)(i % cpad);
  dst[i] = ch < c ? src[r * c + ch] : (bf16_t)0;
}

int conv_wgrad_bf16(const View &v, const void *in_v, int cin, const void *gout_v, int cout, float *gw, void *workspace,
                    hipStream_t s) {
  WgradPlan p = wgrad_plan(v, cin, cout, LGS_BF16);
  char *wsb = reinterpret_cast<char *>(workspace);
  float *partial = reinterpret_cast<float *>(wsb);
  const bf16_t *in = reinterpret_cast<const bf16_t *>(in_v), *go = reinterpret_cast<const bf16_t *>(gout_v);
  const int cin_real = cin, cout_real = cout;
  int64_t pad_off = align256((int64_t)p.S * v.K * p.cin_pad * p.cout_pad * 4);
  if (cin % 8 != 0) {  // e.g. the 3-channel colour input of conv0p1s1: zero-pad rows to 8 channels behind the partials
    const int c8 = (cin + 7) / 8 * 8;
    bf16_t *padded = reinterpret_cast<bf16_t *>(wsb + pad_off);
    int64_t tot = v.n_in * (int64_t)c8;
    if (tot > 0) LGS_KLAUNCH(k_pad_rows_bf16, (unsigned)((tot + 255) / 256), 256, 0, s, in, v.n_in, cin, c8, padded);
    in = padded;
    cin = c8;
    pad_off += align256(tot * 2);
  }
  if (cout % 8 != 0) {  // e.g. the 3-channel offset head of the instance-segmentation model (96 -> 3): pad the gradient rows the
    // same way (the fp32-style fallback took 6.6 ms for this launch at 1.2 M voxels; the reduce kernel drops the padding)
    const int c8 = (cout + 7) / 8 * 8;
    bf16_t *padded = reinterpret_cast<bf16_t *>(wsb + pad_off);
    int64_t tot = v.n_out * (int64_t)c8;
    if (tot > 0) LGS_KLAUNCH(k_pad_rows_bf16, (unsigned)((tot + 255) / 256), 256, 0, s, go, v.n_out, cout, c8, padded);
    go = padded;
    cout = c8;
  }
  // every (slot, k, ci, co) element of `partial` is written exactly once by the wave that owns it
#define LGS_WG(A, B) if (p.nci == A && p.nco == B) { launch_wgrad_bf16<A, B>(v, p, in, cin, go, cout, partial, s); } else
  LGS_WG(1, 1) LGS_WG(1, 2) LGS_WG(1, 3) LGS_WG(1, 4)
  LGS_WG(2, 1) LGS_WG(2, 2) LGS_WG(2, 3) LGS_WG(2, 4)
  LGS_WG(3, 1) LGS_WG(3, 2) LGS_WG(3, 3) LGS_WG(3, 4)
  LGS_WG(4, 1) LGS_WG(4, 2) LGS_WG(4, 3)
  { LGS_REQUIRE(false, "bf16 wgrad: no kernel instance for this tile"); }
#undef LGS_WG
  int64_t total = (int64_t)v.K * cin_real * ((cout_real + 3) / 4);
  LGS_KLAUNCH(k_wgrad_reduce, (unsigned)((total + 255) / 256), 256, 0, s, partial, p.S, v.K, p.cin_pad, p.cout_pad, cin_real,
                     cout_real, gw);
  LGS_HIP(hipGetLastError());
  return 0;
}

__global__ void k_pad_rows_f32(const float *__restrict__ src, int64_t n, int c, int cpad, float *__restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * cpad) return;
  int64_t r = i / cpad;
  int ch = (int)(i % cpad);
  dst[i] = ch < c ? src[r * c + ch] : 0.f;
}

template <typename T>
int conv_wgrad_f32path(const View &v, const void *in_v, int cin, const void *gout_v, int cout, float *gw, void *workspace,
                       hipStream_t s) {
  WgradPlan p = wgrad_plan(v, cin, cout, LGS_F32);
  float *partial = reinterpret_cast<float *>(workspace);
  const T *in = reinterpret_cast<const T *>(in_v);
  const T *go = reinterpret_cast<const T *>(gout_v);
  const int cin_gw = cin;                 // rows of gw[k] (the reduction drops the padding)
  if constexpr (sizeof(T) == 4) {
    // the network's first convolution (3 colour channels; round 6): its weight gradient is the LAST launch of the backward pass,
    // `finalize` waits for exactly it.  Rows padded to 4 channels (16 bytes) go through the staged kernels instead of
    // k_wgrad_f32's one 4-byte load per lane and operand (1.39 ms at 1.2 M voxels)
    if (f32_pad_bytes(v, cin, cout) > 0) {
      const int c4 = (cin + 3) / 4 * 4;
      float *padded = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                                align256((int64_t)p.S * v.K * p.cin_pad * p.cout_pad * 4) + 256);
      const int64_t tot = v.n_in * (int64_t)c4;
      if (tot > 0) LGS_KLAUNCH(k_pad_rows_f32, (unsigned)((tot + 255) / 256), 256, 0, s, reinterpret_cast<const float *>(in), v.n_in, cin, c4, padded);
      in = reinterpret_cast<const T *>(padded);
      cin = c4;
    }
  }
  int n_cot = p.cout_pad / (32 * p.ncb);
  int n_cig = (p.cin_pad / 32 + 3) / 4;
  dim3 grid((unsigned)p.S, (unsigned)v.K, (unsigned)(n_cot * n_cig));
  bool staged = false;
  if constexpr (sizeof(T) == 4) {
    const float *fi = reinterpret_cast<const float *>(in), *fg = reinterpret_cast<const float *>(go);
    // rows off the 16-byte grid (the 3-channel input layer: 1.99 vs 1.70 ms staged element by element, odd head widths) keep k_wgrad_f32
    if (tune(T_WGRAD_F32_LDS) != 0 && cin % 4 == 0 && cout % 4 == 0 && (reinterpret_cast<uintptr_t>(fi) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(fg) & 15u) == 0) {
      staged = true;
#define LGS_WF(KERNEL)                                                                                                          \
      switch (p.ncb) {                                                                                                         \
        case 4: LGS_KLAUNCH((KERNEL<4>), grid, 256, 0, s, v, fi, cin, fg, cout, p.cin_pad, p.cout_pad, p.span, partial); break;  \
        case 3: LGS_KLAUNCH((KERNEL<3>), grid, 256, 0, s, v, fi, cin, fg, cout, p.cin_pad, p.cout_pad, p.span, partial); break;  \
        case 2: LGS_KLAUNCH((KERNEL<2>), grid, 256, 0, s, v, fi, cin, fg, cout, p.cin_pad, p.cout_pad, p.span, partial); break;  \
        default: LGS_KLAUNCH((KERNEL<1>), grid, 256, 0, s, v, fi, cin, fg, cout, p.cin_pad, p.cout_pad, p.span, partial); break; \
      }
      // 2: bf16-split products everywhere; 3: only where they are also faster stand-alone (>= 96 input channels, <= 128 outputs)
      const int64_t mode = tune(T_WGRAD_F32_LDS);
      const bool split = mode == 2 || (mode >= 3 && cin >= 96 && cout <= 128);
      if (split) { LGS_WF(k_wgrad_f32s_lds) } else { LGS_WF(k_wgrad_f32_lds) }
#undef LGS_WF
    }
  }
  if (!staged) switch (p.ncb) {
    case 4: LGS_KLAUNCH((k_wgrad_f32<T, 4>), grid, 256, 0, s, v, in, cin, go, cout, p.cin_pad, p.cout_pad, p.span, partial); break;
    case 3: LGS_KLAUNCH((k_wgrad_f32<T, 3>), grid, 256, 0, s, v, in, cin, go, cout, p.cin_pad, p.cout_pad, p.span, partial); break;
    case 2: LGS_KLAUNCH((k_wgrad_f32<T, 2>), grid, 256, 0, s, v, in, cin, go, cout, p.cin_pad, p.cout_pad, p.span, partial); break;
    default: LGS_KLAUNCH((k_wgrad_f32<T, 1>), grid, 256, 0, s, v, in, cin, go, cout, p.cin_pad, p.cout_pad, p.span, partial); break;
  }
  int64_t total = (int64_t)v.K * cin_gw * ((cout + 3) / 4);
  LGS_KLAUNCH(k_wgrad_reduce, (unsigned)((total + 255) / 256), 256, 0, s, partial, p.S, v.K, p.cin_pad, p.cout_pad, cin_gw,
                     cout, gw);
  LGS_HIP(hipGetLastError());
  return 0;
}

}  // namespace lgs

using namespace lgs;

namespace lgs {
// ---- gradient of the CLIP text-anchor loss w.r.t. the (normalised) anchors: d/dT^ = G^T F^, with G[n, a] the 4-sparse upstream
// gradient dL/dS[n, a] (-g_dpos at the voxel's class, -g_dneg / K at each of its K negatives; zero rows for ignored voxels).
// That is the weight gradient of the 1x1 "convolution" S = F^ T^^T, so it runs on the weight-gradient kernels: the coefficient
// rows (already scaled by 1/|f_n|, so that the RAW features are the other operand) are materialised once in the feature dtype
// and the identity-map launch of k_wgrad_bf16 / k_wgrad_f32 contracts them with the features (fixed summation order).
template <typename T>
__global__ void k_clip_coef_rows(const int64_t *__restrict__ labels, const int64_t *__restrict__ neg, int k_neg, int64_t ignore,
                                 const float *__restrict__ inv_norm, const float *__restrict__ g_dpos, const float *__restrict__ g_dneg,
                                 int64_t n, int n_anchor, int a_ld, T *__restrict__ G) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t lab = labels[r];
  if (lab == ignore || lab < 0 || lab >= n_anchor) return;          // the row stays zero (memset)
  const float inv = inv_norm[r];
  const float gp = g_dpos ? -g_dpos[r] * inv : 0.f;
  const float gn = g_dneg ? -g_dneg[r] * inv / (float)k_neg : 0.f;
  T *row = G + r * (int64_t)a_ld;
  // <= 8 entries; an anchor that occurs more than once (a negative drawn twice, or equal to the class) gets the sum
  int64_t idx[8]; float val[8]; int m = 0;
  idx[m] = lab; val[m] = gp; ++m;
  for (int j = 0; j < k_neg && j < 7; ++j) {
    const int64_t a = neg[r * k_neg + j];
    if (a < 0 || a >= n_anchor) continue;
    int q = 0;
    for (; q < m; ++q) if (idx[q] == a) break;
    if (q < m) val[q] += gn; else { idx[m] = a; val[m] = gn; ++m; }
  }
  for (int q = 0; q < m; ++q) {
    if constexpr (sizeof(T) == 4) row[idx[q]] = val[q]; else row[idx[q]] = f32_to_bf16(val[q]);
  }
}
inline View clip_identity_view(int64_t n) {
  View v;
  v.n_pad = pad_rows(n); v.n_out = n; v.n_in = n; v.KS = 1; v.K = 1;
  return v;
}
}  // namespace lgs

extern "C" {

int64_t lgs_clip_anchor_grad_workspace_bytes(int64_t n, int c, int n_anchor, int dtype) {
  const int a8 = (n_anchor + 7) / 8 * 8;
  const View v = clip_identity_view(n);
  const WgradPlan p = wgrad_plan(v, c, a8, dtype);
  return align256((int64_t)p.S * p.cin_pad * p.cout_pad * 4) + align256(n * (int64_t)a8 * esize(dtype)) + 512;
}

int lgs_clip_loss_backward_anchors(const void *feat, int64_t n, int c, int n_anchor, const int64_t *labels, const int64_t *neg,
                                   int k_neg, int64_t ignore_label, const float *inv_norm_f, const float *g_dpos,
                                   const float *g_dneg, float *grad_anchors_t, int dtype, void *workspace, void *stream) {
  LGS_REQUIRE(feat && labels && neg && inv_norm_f && grad_anchors_t && workspace, "lgs_clip_loss_backward_anchors: null argument");
  LGS_REQUIRE(dtype == LGS_F32 || dtype == LGS_BF16, "lgs_clip_loss_backward_anchors: unknown dtype");
  LGS_REQUIRE(c % (dtype == LGS_BF16 ? 8 : 4) == 0 && k_neg >= 1 && k_neg <= 7 && n_anchor >= 1,
              "lgs_clip_loss_backward_anchors: feature dim on the 16-byte grid, 1..7 negatives");
  hipStream_t s = (hipStream_t)stream;
  const int a8 = (n_anchor + 7) / 8 * 8;
  if (n == 0) {
    LGS_HIP(hipMemsetAsync(grad_anchors_t, 0, sizeof(float) * (size_t)c * a8, s));
    return 0;
  }
  const View v = clip_identity_view(n);
  const WgradPlan p = wgrad_plan(v, c, a8, dtype);
  char *ws = reinterpret_cast<char *>(workspace);
  void *G = ws + align256((int64_t)p.S * p.cin_pad * p.cout_pad * 4);
  LGS_HIP(hipMemsetAsync(G, 0, (size_t)n * a8 * esize(dtype), s));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dtype == LGS_F32)
    LGS_KLAUNCH((k_clip_coef_rows<float>), blocks, 256, 0, s, labels, neg, k_neg, ignore_label, inv_norm_f, g_dpos, g_dneg, n,
                       n_anchor, a8, reinterpret_cast<float *>(G));
  else
    LGS_KLAUNCH((k_clip_coef_rows<bf16_t>), blocks, 256, 0, s, labels, neg, k_neg, ignore_label, inv_norm_f, g_dpos, g_dneg, n,
                       n_anchor, a8, reinterpret_cast<bf16_t *>(G));
  LGS_HIP(hipGetLastError());
  // grad_anchors_t[c][a8] = F^T G   (transposed: the caller reads column a as d/dT^_a)
  if (dtype == LGS_F32) return conv_wgrad_f32path<float>(v, feat, c, G, a8, grad_anchors_t, workspace, s);
  return conv_wgrad_bf16(v, feat, c, G, a8, grad_anchors_t, workspace, s);
}

}  // extern "C"

extern "C" {

int lgs_conv_wgrad_supports_stride(const lgs_kmap *km, int transposed, int cin, int cout, int dtype, int in_row_stride) {
  // mirrors the conditions under which conv_wgrad_ps accepts the call (it is the only weight-gradient kernel that reads a
  // column slice of a wider row-major tensor in place)
  if (!km || dtype != LGS_BF16 || !(km->ks == 3 || km->ks == 2) || km->fwd.n_pad == 0 || !ps_enabled()) return 0;
  if (transposed && km->ks == 3) return 0;
  if (in_row_stride <= 0 || in_row_stride == cin) return 1;
  const int cg = transposed ? cout : cin, cs = transposed ? cin : cout;
  const PsPlan p = ps_plan(km->fwd, cg, cs);
  if (!p.ok || (in_row_stride * 2) % 16 != 0) return 0;
  const View &v = km->fwd;
  const uint64_t rows = transposed ? (uint64_t)v.n_out : (uint64_t)v.n_in;
  return rows * (uint64_t)in_row_stride * 2 < 0xfffff000ull ? 1 : 0;
}

int lgs_conv_wgrad(lgs_kmap *km, int transposed, const void *in, int cin, const void *grad_out, int cout,
                   float *grad_weight, int dtype, void *workspace, int in_row_stride, void *stream) {
  LGS_REQUIRE(km && grad_weight && workspace, "lgs_conv_wgrad: null argument");
  LGS_REQUIRE(!(transposed && km->ks == 3), "transposed 3x3x3 convolution is not part of the model family");
  const View &v = transposed ? km->bwd : km->fwd;  // same view as the forward
  View vv = v; vv.mirror = 0;
  hipStream_t s = (hipStream_t)stream;
  if (kmap_wait(km, s)) return 1;
  if (vv.n_pad == 0) {
    LGS_HIP(hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)km->K * cin * cout, s));
    return 0;
  }
  LGS_REQUIRE(dtype == LGS_BF16 || in_row_stride == 0 || in_row_stride == cin, "lgs_conv_wgrad: strided input needs bf16");
  if (dtype == LGS_F32) return conv_wgrad_f32path<float>(vv, in, cin, grad_out, cout, grad_weight, workspace, s);
  if (dtype == LGS_BF16) {
    if (km->fwd.n_pad > 0 && km->ks == 3 && !transposed) {
      bool done = false;
      int rc = conv_wgrad_wide(km->fwd, in, cin, in_row_stride, grad_out, cout, grad_weight, workspace, s, &done);
      if (rc || done) return rc;
    }
    if (km->fwd.n_pad > 0 && (km->ks == 3 || km->ks == 2)) {
      bool done = false;
      int rc = 0;
      if (!transposed && km->ks == 3 && cin % 8 != 0 && cout % 8 == 0 && (in_row_stride == 0 || in_row_stride == cin) && ps_enabled()) {
        // the 3-channel colour input of the network's first convolution: rows zero-padded to 8 channels behind the partial slabs,
        // then the position-stationary kernel (this launch is the LAST of the backward pass: `finalize` waits for exactly it;
        // the pair-list kernel took 0.31 ms at 1.2 M voxels)
        const int c8 = (cin + 7) / 8 * 8;
        // nothing runs beside the last launch of the backward pass: all 32 CUs of every XCD instead of PS_CUS
        const PsPlan pp = ps_plan(km->fwd, c8, cout, 32);
        if (pp.ok) {
          bf16_t *padded = reinterpret_cast<bf16_t *>(reinterpret_cast<char *>(workspace) + align256(pp.partial_bytes) + 256);
          const int64_t tot = km->fwd.n_in * (int64_t)c8;
          if (tot > 0) LGS_KLAUNCH(k_pad_rows_bf16, (unsigned)((tot + 255) / 256), 256, 0, s, reinterpret_cast<const bf16_t *>(in), km->fwd.n_in, cin, c8, padded);
          rc = conv_wgrad_ps(km->fwd, 0, padded, c8, reinterpret_cast<const bf16_t *>(grad_out), cout, grad_weight, workspace, s, &done, 0, cin, 32);
          if (rc || done) return rc;
        }
      }
      rc = conv_wgrad_ps(km->fwd, transposed, reinterpret_cast<const bf16_t *>(in), cin, reinterpret_cast<const bf16_t *>(grad_out),
                         cout, grad_weight, workspace, s, &done, in_row_stride);
      if (rc || done) return rc;
    }
    LGS_REQUIRE(in_row_stride == 0 || in_row_stride == cin, "lgs_conv_wgrad: a strided input is only supported by the position-stationary bf16 kernel");
    return conv_wgrad_bf16(vv, in, cin, grad_out, cout, grad_weight, workspace, s);
  }
  LGS_REQUIRE(false, "lgs_conv_wgrad: unknown dtype");
}

}  // extern "C"
