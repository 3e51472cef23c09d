// lgs_conv_halo.hip -- 3^3 stride-1 sparse convolution forward / dgrad for NARROW channel counts (<= 96 gathered channels per
// input pass, <= 96 output channels per output pass, bf16) on the big maps of levels 0-2, on gfx950.
//
// Serves the same call sites as lgs_conv.hip's k_conv_gather (MinkowskiConvolution forward + autograd dgrad,
//   /root/reference/models/modules/common.py:179-203, models/modules/resnet_block.py:41-57, models/res16unet.py:196-270):
// the 3^3 convolutions 96 -> 96, 128 -> 96, 32 -> 32, 64 -> 64 ... of Res16UNet34C's blocks at tensor strides 1, 2, 4, which are
// 61 % of the compute stream of the benchmarked step.
//
// Why another kernel.  k_conv_gather fetches the neighbour rows of every (offset, 32-row block) with its own wave-wide gather
// instruction straight into MFMA operand registers: 713 gather instructions per 256 positions at level 0 (45 % of them blocks
// without a single neighbour, kept for the static schedule), and a CU retires a gather instruction every ~20 ns whatever it
// carries (tools/probes/gather_probe.hip): the launch sat on the vector-memory ISSUE rate at 33 % of the HBM peak with the
// MFMA pipe 25 % busy.  But a 256-position Morton tile of a surface scan touches only ~380 DISTINCT input rows (p95 ~430):
// its own 256 and a thin halo.  Here
//   * the coordinate manager hands over, per tile, that row list and the kernel map rewritten as 16-bit slots into it
//     (lgs_common.h HaloView, lgs_manager.hip k_build_halo);
//   * a workgroup (4 waves, one per SIMD) stages the tile's rows ONCE in LDS by LDS-DMA -- 16-byte pieces, one instruction per
//     ~5 rows: ~78 instructions per tile instead of 713 -- at a row pitch of (channels x 2 + 16) bytes, so that consecutive
//     slots start 13 / 9 / 5 sixteen-byte units apart (odd: neighbouring slots fall on different banks);
//   * the work of the tile is split over the waves by OFFSET PARITY x ROW HALF: wave (o, r) multiplies the offsets k = o (mod 2)
//     against the four 32-row blocks 2 j + r (parity = rank of k among the tile's active offsets).  The offset's weight block (pre-packed in MFMA fragment order: the image layout
//     k_conv_gather uses) goes straight from L2 into the wave's REGISTERS, one offset ahead (two register sets), and serves
//     all four row blocks: no weights in LDS, no barrier anywhere in the main loop (version 1 staged weight slabs in LDS and
//     crossed a barrier per offset: at one wave per SIMD every latency was exposed, 0.87 vs 0.57 ms at level 0);
//   * the B operands (neighbour rows) are read from LDS through the slot (one ds_read_b128 per lane and 16-channel k-step; a
//     missing neighbour reads the tile's zero row), one row block AHEAD of the MFMAs that consume them, the next offset's
//     slots one offset ahead; 32-row blocks without a neighbour at an offset skip their MFMAs;
//   * at the end the two offset-parity waves of a row half exchange half of their fp32 accumulators through the (now idle)
//     row buffer and each finishes two row blocks: fixed summation order, every output row written once, no atomics;
//   * tiles whose list does not fit the LDS row buffer are processed in list SEGMENTS (slots outside the staged segment read
//     the zero row); tiles with more than kHaloS distinct rows (count = -1) stage the 256 neighbour rows of one offset at a
//     time, like k_conv_wide does.  Both are rare on surface scans and exact.
// Gathered rows wider than 96 channels go in input PASSES over channel slices (128 = 2 x 64, accumulators persist); outputs
// wider than 96 channels in output passes of 64 (the row staging is repeated; 128 accumulator registers per wave stay 192).
#include "lgs_common.h"

#include <type_traits>

namespace lgs {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LGS_HALO_AS3(p) ((__attribute__((address_space(3))) void *)(p))
#define LGS_HALO_VMCNT0() __builtin_amdgcn_s_waitcnt(((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14)))

constexpr int kHaloLds = 160 * 1024;          // LDS of one workgroup = the CU's

template <int NC, int NBP>
struct HaloCfg {
  static constexpr int PITCH = 64 * NC + 16;                       // bytes per staged row
  static constexpr int UPR = PITCH / 16;                           // 16-byte units per row (the last one is padding)
  static constexpr int RED = 4 * 2 * NBP * 16 * 64 * 4;            // accumulator exchange: 4 waves x 2 row blocks x NBP tiles, fp32
  static constexpr int ROWS_MAX = (kHaloLds - 64) / PITCH - 1;
  static constexpr int CAP0 = ROWS_MAX >= kHaloS ? kHaloS : (ROWS_MAX / 32 * 32);
  static constexpr int CAP = CAP0 > 512 ? 512 : CAP0;              // rows per staged segment (512: <= 26 DMA instructions per thread)
  static constexpr int ROWS_B = (CAP + 1) * PITCH;                 // + the zero row
  static constexpr int ITR = (CAP * UPR + 255) / 256;              // DMA instructions per thread for a full segment
  static constexpr int LDS = ROWS_B > RED ? ROWS_B : RED;
  static_assert(LDS <= kHaloLds, "LDS budget");
  static_assert(CAP >= kHaloT, "a tile's own rows must fit one segment");
};

// TRACE (tuning knob HALO_TRACE, debug instance): shader-clock sums of wave 0 of one tile in the middle of the grid:
// [0] row staging (to the barrier)  [1] main loop  [2] accumulator exchange + stores  [3] offsets walked  [4] active (block, offset)
// pairs of the wave  [5] distinct rows of the tile
template <int NC, int NBP, bool TRACE>
__global__ __launch_bounds__(256, 1) void k_conv_halo(HaloView hv, const bf16_t *__restrict__ in, int g_real, int in_ld, int npass_in,
                                                       const u32x4 *__restrict__ wp, int ncp, int nbp, int npass_out,
                                                       bf16_t *__restrict__ out, int o_real, const float *__restrict__ bias, int accum,
                                                       unsigned in_bytes, unsigned w_bytes, unsigned long long *trace) {
  using C = HaloCfg<NC, NBP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vx = lane & 31, h = lane >> 5;
  const int wo = wave & 1, wr = wave >> 1;                          // offset parity, row half
  const View &v = hv.v;

  // XCD-aware order (speed only): workgroup b runs on XCD b % 8; give every XCD a contiguous run of tiles
  int64_t tile;
  {
    const unsigned nt = gridDim.x, xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, q = nt >> 3, r = nt & 7u;
    tile = (int64_t)(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int64_t pos0 = tile * kHaloT;
  uint32_t smask = 0;
#pragma unroll
  for (int g = 0; g < kHaloT / 64; ++g) smask |= v.mask64[pos0 / 64 + g];
  smask = __builtin_amdgcn_readfirstlane(smask);
  const int U = __builtin_amdgcn_readfirstlane(hv.ucount[tile]);
  const bool listed = U >= 0;                                   // false: per-offset staging
  const int nseg = listed ? (U + C::CAP - 1) / C::CAP : 27;
  const bool tr = TRACE && trace != nullptr && blockIdx.x == (gridDim.x / 2 / 8) * 8 && wave == 0;
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tt0 = 0, tt1 = 0;
  if (tr) { tt0 = __builtin_amdgcn_s_memtime(); tacc[5] = (unsigned long long)(U < 0 ? 0 : U); }

  char *l_rows = smem;
  constexpr unsigned kOOB = 0xfffff000u;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wp), 0, (int)w_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)in_ld * 2u;
  // this wave's four 32-row blocks 2 j + wr (the tile's rows are sorted by neighbourhood mask: interleaving gives both halves
  // the same mix), its row in each
  int rowj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rowj[j] = (2 * j + wr) * 32 + vx;

  for (int op = 0; op < npass_out; ++op) {
    f32x16 acc[4][NBP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NBP; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

    for (int ip = 0; ip < npass_in; ++ip) {
      const unsigned ch0 = (unsigned)(ip * NC * 64);                // byte offset of this pass's channel slice inside a row
      const unsigned lim = (unsigned)g_real * 2u;                   // bytes of real channels in a row
      for (int seg = 0; seg < nseg; ++seg) {
        if (!listed && !((smask >> seg) & 1u)) continue;            // workgroup-uniform
        const int sb = listed ? seg * C::CAP : 0;                   // first list slot of this segment
        const int cnt = listed ? min(C::CAP, U - sb) : kHaloT;      // rows staged
        const uint32_t em = listed ? smask : (1u << seg);           // offsets multiplied against this staging
        __syncthreads();                                            // everybody is done with the previous use of the row buffer
        // ---- stage the rows: unit e = 16-byte piece e % UPR of list row e / UPR; LDS side linear in e
        {
          // the zero row (slot CAP): what a missing neighbour, a slot outside the staged segment and a padding position read
          if (tid < C::UPR) *reinterpret_cast<u32x4 *>(l_rows + C::CAP * C::PITCH + tid * 16) = u32x4{0u, 0u, 0u, 0u};
          const int32_t *src = listed ? hv.urows + tile * kHaloS + sb : v.nbr + (int64_t)seg * v.n_pad + pos0;
          const int total = cnt * C::UPR;
          // unconditional loads, clamped to the addressable part of the list (its stride is kHaloS entries; a tile has kHaloT
          // own rows): they do not depend on the tile's row count, so they are in flight together with it
          const int rmax = listed ? (kHaloS - sb - 1) : (kHaloT - 1);
          int32_t rid[C::ITR];
#pragma unroll
          for (int it = 0; it < C::ITR; ++it) {
            const int e = it * 256 + tid, r = min(e / C::UPR, rmax);
            rid[it] = src[r];
          }
#pragma unroll
          for (int it = 0; it < C::ITR; ++it) {
            const int e = it * 256 + tid;
            if (e < total) {                                        // (inactive lanes of a DMA instruction write nothing)
              const int r = e / C::UPR, piece = e - r * C::UPR;
              const unsigned cb = ch0 + (unsigned)piece * 16u;
              const bool ok = rid[it] >= 0 && piece < C::UPR - 1 && cb + 16u <= lim;
              const unsigned off = ok ? (unsigned)rid[it] * row_bytes + cb : kOOB;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, LGS_HALO_AS3(l_rows + (it * 256 + wave * 64) * 16), 16, off, 0, 0, 0);
            }
          }
        }
        // ---- main loop of this wave: the offsets of its parity, ascending; no barrier, nothing shared but the read-only rows
        // the active offsets alternate between the two parity waves by RANK (not by k: the tile's offset set is arbitrary)
        uint32_t rem = 0;
        {
          uint32_t m = em;
          int rank = 0;
          while (m != 0) {
            const uint32_t b = m & (0u - m);
            if ((rank & 1) == wo) rem |= b;
            m ^= b;
            ++rank;
          }
        }
        // (a wave without an offset still takes part in the staging barriers)
        {
          // slots of the wave's four rows at offset k -> LDS byte address of each row (the zero row if none) + activity bits
          auto load_slots = [&](int k, unsigned (&sl)[4]) __attribute__((always_inline)) {
            if (listed) {
#pragma unroll
              for (int j = 0; j < 4; ++j) sl[j] = k >= 0 ? (unsigned)hv.lnbr[(int64_t)k * v.n_pad + pos0 + rowj[j]] : 0xffffu;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) sl[j] = (k >= 0 && v.nbr[(int64_t)k * v.n_pad + pos0 + rowj[j]] >= 0) ? (unsigned)rowj[j] : 0xffffu;
            }
          };
          auto resolve = [&](const unsigned (&sl)[4], unsigned (&ad)[4], uint32_t &act) __attribute__((always_inline)) {
            act = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned q = sl[j] - (unsigned)sb;                        // 0xffff - sb stays >= cnt
              const bool ok = q < (unsigned)cnt;
              if (__ballot(ok) != 0ull) act |= 1u << j;
              ad[j] = (ok ? q : (unsigned)C::CAP) * (unsigned)C::PITCH + (unsigned)h * 32u;
            }
          };
          auto wload = [&](int k, u32x4 (&w)[NC][NBP][2]) __attribute__((always_inline)) {
            // packed image [k][chunk][block][t][lane] x 16 B: fragment (c, nb, t) of this input / output pass
            const unsigned base = k >= 0 ? (unsigned)((((int64_t)k * ncp + ip * NC) * nbp + op * NBP) * 2048) + (unsigned)lane * 16u : kOOB;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int nb = 0; nb < NBP; ++nb)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                  w[c][nb][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, base, ((c * nbp + nb) * 2 + t) * 1024, 0);   // fragment offset: scalar
          };
          auto bread = [&](unsigned ad, u32x4 (&f)[NC][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int t = 0; t < 2; ++t) f[c][t] = *reinterpret_cast<const u32x4 *>(l_rows + ad + 64 * c + 16 * t);
          };

          u32x4 w0[NC][NBP][2], w1[NC][NBP][2], fb[2][NC][2];
          unsigned sl[4], ad[4], adn[4];
          uint32_t act = 0, actn = 0;
          const bool any = rem != 0;
          const int k = any ? __builtin_ctz(rem) : -1;
          rem &= rem - 1;
          load_slots(k, sl);                                        // the first offset's slots and weights travel with the row DMA
          wload(k, w0);
          LGS_HALO_VMCNT0();                                        // this wave's row pieces have landed
          __syncthreads();                                          // everybody's have
          if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[0] += tt1 - tt0; tt0 = tt1; }
          resolve(sl, ad, act);
          bread(ad[0], fb[0]);
          // one offset, block by block.  Everything the NEXT block / offset needs is issued in the shadow of this block's MFMAs: a
          // quarter of the next offset's weight fragments (L2 -> registers), the six B fragments of the next block (LDS), and at
          // the last block the next offset's slots resolved into row addresses.  An MFMA occupies the SIMD's matrix pipe for 32
          // cycles and the wave waits at the next one: one memory instruction per MFMA issues for free there, while 18 weight
          // loads in a burst in front of the MFMAs cost ~1000 dead cycles per offset (first version of this loop).
          constexpr int NF = NC * NBP * 2;                           // weight fragments per offset
          auto wload_part = [&](int kn, int j, u32x4 (&w)[NC][NBP][2]) __attribute__((always_inline)) {
            const unsigned base = kn >= 0 ? (unsigned)((((int64_t)kn * ncp + ip * NC) * nbp + op * NBP) * 2048) + (unsigned)lane * 16u : kOOB;
#pragma unroll
            for (int f = (NF * j) / 4; f < (NF * (j + 1)) / 4; ++f) {
              const int c = f / (NBP * 2), nb = (f / 2) % NBP, t = f & 1;
              w[c][nb][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, base, ((c * nbp + nb) * 2 + t) * 1024, 0);
            }
          };
          auto step = [&](u32x4 (&wc)[NC][NBP][2], u32x4 (&wn)[NC][NBP][2]) __attribute__((always_inline)) -> bool {
            const int kn = rem != 0 ? __builtin_ctz(rem) : -1;
            rem &= rem - 1;
            load_slots(kn, sl);
            if (tr) { tacc[3] += 1; tacc[4] += (unsigned long long)__builtin_popcount(act); }
            auto block = [&](auto JC) __attribute__((always_inline)) {
              constexpr int j = decltype(JC)::value;
              auto prefetch = [&]() __attribute__((always_inline)) {
                wload_part(kn, j, wn);
                if constexpr (j < 3) {
                  bread(ad[j + 1], fb[(j + 1) & 1]);
                } else {
                  resolve(sl, adn, actn);
                  bread(adn[0], fb[0]);
                }
              };
              // NO branch around a block without neighbours at this offset: its B reads hit the zero row and its MFMAs add zeros
              // (~27 % of the (block, offset) pairs of a surface tile).  Skipping them needs the prefetches in both arms of a
              // branch -- hipcc then spills hundreds of registers -- or a burst of 18 weight loads in front of the MFMAs
              // (measured: 1025 cycles per active pair against 576 of MFMA time).  Straight-line code keeps the matrix pipe fed.
              __builtin_amdgcn_sched_barrier(0);
              prefetch();
#pragma unroll
              for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                  for (int nb = 0; nb < NBP; ++nb)
                    acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[c][nb][t]),
                                                                         __builtin_bit_cast(bf16x8, fb[j & 1][c][t]), acc[j][nb], 0, 0, 0);
              // order: MFMA, then one weight load / one LDS read per MFMA until they are all out, then the remaining MFMAs
              constexpr int NW = (NF * (j + 1)) / 4 - (NF * j) / 4, NR = 2 * NC, NM = NF;
#pragma unroll
              for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < NW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
            };
            block(std::integral_constant<int, 0>{});
            block(std::integral_constant<int, 1>{});
            block(std::integral_constant<int, 2>{});
            block(std::integral_constant<int, 3>{});
#pragma unroll
            for (int j = 0; j < 4; ++j) ad[j] = adn[j];
            act = actn;
            return kn >= 0;
          };
          if (any)
            for (;;) {
              if (!step(w0, w1)) break;
              if (!step(w1, w0)) break;
            }
        }
        if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[1] += tt1 - tt0; tt0 = tt1; }
      }
    }

    // ---- the two offset-parity waves of a row half exchange accumulators: parity 0 finishes the half's blocks j = 0, 1,
    // parity 1 finishes j = 2, 3; each hands the other pair over through LDS.  [wave][jj][nb][q][lane] x 16 B
    __syncthreads();                                                // the row buffer is idle
    {
      float *xw = reinterpret_cast<float *>(smem + wave * (C::RED / 4));
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int nb = 0; nb < NBP; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 t4;
            t4.x = wo ? acc[jj][nb][4 * q + 0] : acc[jj + 2][nb][4 * q + 0];
            t4.y = wo ? acc[jj][nb][4 * q + 1] : acc[jj + 2][nb][4 * q + 1];
            t4.z = wo ? acc[jj][nb][4 * q + 2] : acc[jj + 2][nb][4 * q + 2];
            t4.w = wo ? acc[jj][nb][4 * q + 3] : acc[jj + 2][nb][4 * q + 3];
            *reinterpret_cast<float4 *>(xw + ((((jj * NBP + nb) * 4 + q) * 64 + lane) * 4)) = t4;
          }
    }
    __syncthreads();
    {
      const float *xr = reinterpret_cast<const float *>(smem + (wave ^ 1) * (C::RED / 4));
      // ---- epilogue: lane (voxel vx, half h) owns channels nb*32 + 8q + 4h + {0..3} of its rows
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int64_t p = pos0 + (wo ? rowj[jj + 2] : rowj[jj]);
        const int32_t orow = v.out_row[p];
        bf16_t *dst = out + (int64_t)(orow < 0 ? 0 : orow) * o_real;
#pragma unroll
        for (int nb = 0; nb < NBP; ++nb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4 *>(xr + ((((jj * NBP + nb) * 4 + q) * 64 + lane) * 4));
            const int c0 = (op * NBP + nb) * 32 + 8 * q + 4 * h;
            if (orow < 0 || c0 >= o_real) continue;
            // fixed order: even offsets' sum + odd offsets' sum
            float o0 = wo ? t4.x + acc[jj + 2][nb][4 * q + 0] : acc[jj][nb][4 * q + 0] + t4.x;
            float o1 = wo ? t4.y + acc[jj + 2][nb][4 * q + 1] : acc[jj][nb][4 * q + 1] + t4.y;
            float o2 = wo ? t4.z + acc[jj + 2][nb][4 * q + 2] : acc[jj][nb][4 * q + 2] + t4.z;
            float o3 = wo ? t4.w + acc[jj + 2][nb][4 * q + 3] : acc[jj][nb][4 * q + 3] + t4.w;
            if (bias) { o0 += bias[c0]; o1 += bias[c0 + 1]; o2 += bias[c0 + 2]; o3 += bias[c0 + 3]; }
            if (accum) {   // kernel-uniform: rounded exactly like "store the result, then add the two tensors" (lgs_conv_dgrad_accumulate)
              const uint2 t = *reinterpret_cast<const uint2 *>(dst + c0);
              o0 = bf16_to_f32(f32_to_bf16(o0)) + __uint_as_float(t.x << 16);
              o1 = bf16_to_f32(f32_to_bf16(o1)) + __uint_as_float(t.x & 0xffff0000u);
              o2 = bf16_to_f32(f32_to_bf16(o2)) + __uint_as_float(t.y << 16);
              o3 = bf16_to_f32(f32_to_bf16(o3)) + __uint_as_float(t.y & 0xffff0000u);
            }
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);
            pk.y = (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16);
            *reinterpret_cast<uint2 *>(dst + c0) = pk;
          }
        }
      }
    }
    if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[2] += tt1 - tt0; tt0 = tt1; }
  }
  if (tr && lane == 0)
    for (int i = 0; i < 6; ++i) trace[i] = tacc[i];
}

// ------------------------------------------------------------------------------------------------ host side
namespace {
struct HaloShape { int nc, npass_in, nbp, npass_out; };
inline bool halo_shape(int g_real, int o_real, HaloShape *hs) {
  if (g_real % 8 != 0 || o_real % 4 != 0 || g_real < 32 || o_real < 8 || o_real > 128) return false;
  const int gc = pad32(g_real) / 32, nb = pad32(o_real) / 32;
  int nc, npi;
  if (gc <= 3) { nc = gc; npi = 1; }
  else if (gc == 4) { nc = 2; npi = 2; }
  else if (gc == 6) { nc = 3; npi = 2; }
  else return false;
  const int nbp = nb == 4 ? 2 : nb, npo = nb == 4 ? 2 : 1;
  // instantiated (NC, NBP) pairs: the shapes of Res16UNet34C / 14A / 18 at levels 0-2 in both directions
  const bool have = (nc == 1 && (nbp == 1 || nbp == 2)) || (nc == 2 && nbp >= 1 && nbp <= 3) || (nc == 3 && (nbp == 2 || nbp == 3));
  if (!have) return false;
  hs->nc = nc; hs->npass_in = npi; hs->nbp = nbp; hs->npass_out = npo;
  return true;
}
}  // namespace

bool conv_halo_supported(const HaloView &hv, int g_real, int o_real, int K) {
  HaloShape hs;
  return hv.ok && K == 27 && hv.v.n_pad > 0 && tune(T_HALO) != 0 && halo_shape(g_real, o_real, &hs);
}

int64_t conv_halo_pack_layout(int g_real, int o_real, int *ncp, int *nbp) {
  HaloShape hs;
  if (!halo_shape(g_real, o_real, &hs)) return 0;
  *ncp = hs.nc * hs.npass_in; *nbp = hs.nbp * hs.npass_out;
  return (int64_t)27 * (*ncp) * (*nbp) * 2 * 64;     // 16-byte units
}

template <int NC, int NBP>
static int launch_halo_t(const HaloView &hv, const void *in, int g_real, int in_ld, const HaloShape &hs, const void *wp, int ncp, int nbp, void *out,
                         int o_real, const float *bias, int accum, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
  using C = HaloCfg<NC, NBP>;
  static bool attr_set = false;
  if (!attr_set) {
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_halo<NC, NBP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_halo<NC, NBP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    attr_set = true;
  }
  const unsigned nt = (unsigned)(hv.v.n_pad / kHaloT);
  if (tune(T_HALO_TRACE) != 0) {
    static unsigned long long *trace = nullptr;
    if (!trace) LGS_HIP(hipMalloc(&trace, 8 * sizeof(unsigned long long)));
    LGS_HIP(hipMemsetAsync(trace, 0, 8 * sizeof(unsigned long long), s));
    LGS_KLAUNCH((k_conv_halo<NC, NBP, true>), dim3(nt), dim3(256), C::LDS, s, hv, reinterpret_cast<const bf16_t *>(in), g_real, in_ld, hs.npass_in,
                reinterpret_cast<const u32x4 *>(wp), ncp, nbp, hs.npass_out, reinterpret_cast<bf16_t *>(out), o_real, bias, accum, in_bytes, w_bytes, trace);
    unsigned long long h[8];
    LGS_HIP(hipStreamSynchronize(s));
    LGS_HIP(hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[k_conv_halo trace] %d->%d NC %d NBP %d: tile of %llu distinct rows, wave 0 walked %llu offsets with %llu active (block, offset) pairs; "
            "cycles: staging %llu  main loop %llu  exchange + stores %llu\n", g_real, o_real, NC, NBP, h[5], h[3], h[4], h[0], h[1], h[2]);
    return 0;
  }
  LGS_KLAUNCH((k_conv_halo<NC, NBP, false>), dim3(nt), dim3(256), C::LDS, s, hv, reinterpret_cast<const bf16_t *>(in), g_real, in_ld, hs.npass_in,
              reinterpret_cast<const u32x4 *>(wp), ncp, nbp, hs.npass_out, reinterpret_cast<bf16_t *>(out), o_real, bias, accum, in_bytes, w_bytes,
              (unsigned long long *)nullptr);
  LGS_HIP(hipGetLastError());
  return 0;
}

int launch_conv_halo(const HaloView &hv, int mirror, const void *in, int g_real, int in_ld, const void *wp, int ncp, int nbp,
                     void *out, int o_real, const float *bias, int accum, hipStream_t s) {
  (void)mirror;                                      // the dgrad mirroring (K - 1 - k) is folded into the weight packing
  HaloShape hs;
  LGS_REQUIRE(halo_shape(g_real, o_real, &hs) && ncp == hs.nc * hs.npass_in && nbp == hs.nbp * hs.npass_out,
              "halo conv: shape / packed-image mismatch (internal error)");
  const int ld = in_ld > 0 ? in_ld : g_real;
  const uint64_t in_bytes64 = (uint64_t)hv.v.n_in * (uint64_t)ld * 2, w_bytes64 = (uint64_t)27 * ncp * nbp * 2048;
  LGS_REQUIRE(in_bytes64 < 0xfffff000ull && w_bytes64 < 0xfffff000ull && (ld * 2) % 16 == 0,
              "sparse conv: a feature or weight tensor of 4 GiB or more is beyond the 32-bit buffer-descriptor path");
#define LGS_HALO_CASE(NCV, NBV)                                                                                              \
  if (hs.nc == NCV && hs.nbp == NBV)                                                                                         \
    return launch_halo_t<NCV, NBV>(hv, in, g_real, ld, hs, wp, ncp, nbp, out, o_real, bias, accum, (unsigned)in_bytes64, (unsigned)w_bytes64, s);
  LGS_HALO_CASE(1, 1) LGS_HALO_CASE(1, 2)
  LGS_HALO_CASE(2, 1) LGS_HALO_CASE(2, 2) LGS_HALO_CASE(2, 3)
  LGS_HALO_CASE(3, 2) LGS_HALO_CASE(3, 3)
#undef LGS_HALO_CASE
  LGS_REQUIRE(false, "halo conv: no kernel instance for this shape (internal error)");
}

}  // namespace lgs
