// lgs_conv_halo.hip -- 3^3 stride-1 sparse convolution forward / dgrad for NARROW channel counts (<= 128 gathered channels per
// pass, <= 128 output channels, bf16) on the big maps of levels 0-2, on gfx950.
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
//   * a workgroup (4 waves, one per SIMD, 64 positions each) stages the tile's rows ONCE in LDS by LDS-DMA -- 16-byte pieces,
//     one instruction per ~5 rows: ~78 instructions per tile instead of 713 -- at a row pitch of (channels x 2 + 16) bytes, so
//     that consecutive slots start 13 / 9 / 5 sixteen-byte units apart (odd: neighbouring slots fall on different banks);
//   * per offset the B operands (neighbour rows) are read from LDS through the slot (one ds_read_b128 per lane and 16-channel
//     k-step; a missing neighbour reads the tile's zero row), the A operands (the offset's weight block, pre-packed in MFMA
//     fragment order: the image k_conv_gather uses) from a double-buffered LDS slab that is refilled through registers one
//     slab ahead; fp32 accumulation in registers; every output row is written once (deterministic, no atomics);
//   * positions are clustered by neighbourhood mask INSIDE the tile, a wave owns the tile's 32-row blocks w and 7 - w (the
//     two ends of the sorted order meet in one wave: waves see similar numbers of active (block, offset) pairs between the
//     per-slab barriers), blocks without a neighbour at an offset are skipped;
//   * tiles whose list does not fit the LDS row buffer are processed in list SEGMENTS (slots outside the staged segment read
//     the zero row); tiles with more than kHaloS distinct rows (count = -1) stage the 256 neighbour rows of one offset at a
//     time, like k_conv_wide does.  Both are rare on surface scans and exact.
// Gathered rows wider than 96 channels go in PASSES over channel slices (128 = 2 x 64), accumulators persist.
#include "lgs_common.h"

#include <type_traits>

namespace lgs {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LGS_HALO_AS3(p) ((__attribute__((address_space(3))) void *)(p))
#define LGS_HALO_VMCNT0() __builtin_amdgcn_s_waitcnt(((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14)))

constexpr int kHaloLds = 160 * 1024;          // LDS of one workgroup = the CU's
constexpr int kHaloWBudget = 24576;           // bytes of ONE weight slab buffer (two of them)
constexpr int kHaloTab = 27 * kHaloT * 2;     // slot table of the tile

template <int NC, int NB>
struct HaloCfg {
  static constexpr int PITCH = 64 * NC + 16;                       // bytes per staged row
  static constexpr int UPR = PITCH / 16;                           // 16-byte units per row (the last one is padding)
  static constexpr int WK = NC * NB * 2048;                        // bytes of one offset's weight block (one pass)
  static constexpr int G = (kHaloWBudget / WK) < 1 ? 1 : ((kHaloWBudget / WK) > 27 ? 27 : (kHaloWBudget / WK));   // offsets per slab
  static constexpr int WSLAB = G * WK;
  static constexpr int WRN = (WSLAB / 16 + 255) / 256;             // 16-byte staging registers per thread
  static constexpr int ROWS_MAX = (kHaloLds - 2 * WSLAB - kHaloTab - 64) / PITCH - 1;
  static constexpr int CAP = ROWS_MAX >= kHaloS ? kHaloS : (ROWS_MAX / 32 * 32);      // rows per staged segment
  static constexpr int ROWS_B = (CAP + 1) * PITCH;                 // + the zero row
  static constexpr int ITR = (CAP * UPR + 255) / 256;              // DMA instructions per thread for a full segment
  static constexpr int O_ROWS = 0, O_W = ROWS_B, O_TAB = O_W + 2 * WSLAB, O_KL = O_TAB + kHaloTab, LDS = O_KL + 32;
  static_assert(LDS <= kHaloLds, "LDS budget");
  static_assert(CAP >= kHaloT, "a tile's own rows must fit one segment");
};

// TRACE (tuning knob HALO_TRACE, debug instance): shader-clock sums of wave 0 of one tile in the middle of the grid:
// [0] prologue (table + row staging + first slab, to the first barrier)  [1] weight prefetch issue  [2] multiply  [3] slab end
// (weight store + barrier)  [4] epilogue  [5] slabs  [6] active (block, offset) pairs of the wave  [7] distinct rows
template <int NC, int NB, bool TRACE>
__global__ __launch_bounds__(256, 1) void k_conv_halo(HaloView hv, const bf16_t *__restrict__ in, int g_real, int in_ld, int npass,
                                                       const u32x4 *__restrict__ wp, int ncp, bf16_t *__restrict__ out, int o_real,
                                                       const float *__restrict__ bias, int accum, unsigned in_bytes, unsigned w_bytes,
                                                       unsigned long long *trace) {
  using C = HaloCfg<NC, NB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vx = lane & 31, h = lane >> 5;
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
  const bool tr = TRACE && trace != nullptr && blockIdx.x == (gridDim.x / 2 / 8) * 8 && wave == 0;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tt0 = 0, tt1 = 0;
  if (tr) { tt0 = __builtin_amdgcn_s_memtime(); tacc[7] = (unsigned long long)(U < 0 ? 0 : U); }
  const bool listed = U >= 0;                                   // false: per-offset staging
  const int nseg = listed ? (U + C::CAP - 1) / C::CAP : 27;

  char *l_rows = smem + C::O_ROWS;
  uint16_t *l_tab = reinterpret_cast<uint16_t *>(smem + C::O_TAB);
  constexpr unsigned kOOB = 0xfffff000u;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wp), 0, (int)w_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)in_ld * 2u;

  // this wave's two 32-row blocks: w and 7 - w of the tile (sorted by neighbourhood mask)
  const int blk0 = wave, blk1 = 7 - wave;
  const int row0 = blk0 * 32 + vx, row1 = blk1 * 32 + vx;

  f32x16 acc[2][NB];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;

  // the zero row (slot CAP): what a missing neighbour, a slot outside the staged segment and a padding position read
  if (tid < C::UPR) *reinterpret_cast<u32x4 *>(l_rows + C::CAP * C::PITCH + tid * 16) = u32x4{0u, 0u, 0u, 0u};
  // slot table of the tile (listed tiles): offset-major [27][256] uint16
  if (listed) {
    uint16_t tv[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) tv[k] = ((smask >> k) & 1u) ? hv.lnbr[(int64_t)k * v.n_pad + pos0 + tid] : (uint16_t)0xffffu;
#pragma unroll
    for (int k = 0; k < 27; ++k) l_tab[k * kHaloT + tid] = tv[k];
  }

  for (int pass = 0; pass < npass; ++pass) {
    const unsigned ch0 = (unsigned)(pass * NC * 64);                // byte offset of this pass's channel slice inside a row
    const unsigned lim = (unsigned)g_real * 2u;                     // bytes of real channels in a row
    for (int seg = 0; seg < nseg; ++seg) {
      if (!listed && !((smask >> seg) & 1u)) continue;              // workgroup-uniform
      const int sb = listed ? seg * C::CAP : 0;                     // first list slot of this segment
      const int cnt = listed ? min(C::CAP, U - sb) : kHaloT;        // rows staged
      const uint32_t em = listed ? smask : (1u << seg);             // offsets multiplied against this staging
      __syncthreads();                                              // everybody is done with the previous staging (rows, table, klist)
      // ---- stage the rows: unit e = 16-byte piece e % UPR of list row e / UPR; LDS side linear in e
      {
        const int32_t *src = listed ? hv.urows + tile * kHaloS + sb : v.nbr + (int64_t)seg * v.n_pad + pos0;
        const int total = cnt * C::UPR;
        int32_t rid[C::ITR];
#pragma unroll
        for (int it = 0; it < C::ITR; ++it) {
          const int e = it * 256 + tid, r = min(e / C::UPR, cnt - 1);   // unconditional, clamped: a conditional load is fenced with vmcnt(0)
          rid[it] = src[r];
        }
        if (!listed) {                                              // slots of the one offset: the position itself
          const int32_t mine = v.nbr[(int64_t)seg * v.n_pad + pos0 + tid];
          l_tab[seg * kHaloT + tid] = mine >= 0 ? (uint16_t)tid : (uint16_t)0xffffu;
        }
#pragma unroll
        for (int it = 0; it < C::ITR; ++it) {
          const int e = it * 256 + tid;
          if (e < total) {                                          // (inactive lanes of a DMA instruction write nothing)
            const int r = e / C::UPR, piece = e - r * C::UPR;
            const unsigned cb = ch0 + (unsigned)piece * 16u;
            const bool ok = rid[it] >= 0 && piece < C::UPR - 1 && cb + 16u <= lim;
            const unsigned off = ok ? (unsigned)rid[it] * row_bytes + cb : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, LGS_HALO_AS3(l_rows + (it * 256 + wave * 64) * 16), 16, off, 0, 0, 0);
          }
        }
      }
      // ---- active offsets of this staging, in ascending order: two scalar bit iterators (weight side one slab ahead)
      const int nk = __builtin_popcount(em);
      uint32_t wrem = em, crem = em;
      // ---- weights through registers: slab = the next G offsets of `wrem`; unit e of the slab buffer = piece e % (WK / 16)
      // of the slab's offset e / (WK / 16)
      u32x4 wreg[C::WRN];
      auto wload = [&]() __attribute__((always_inline)) {
        int ks[C::G];
#pragma unroll
        for (int j = 0; j < C::G; ++j) {
          ks[j] = wrem ? __builtin_ctz(wrem) : -1;
          wrem &= wrem - 1;
        }
#pragma unroll
        for (int i = 0; i < C::WRN; ++i) {
          const int e = tid + i * 256, oi = e / (C::WK / 16), wi = e - oi * (C::WK / 16);
          int k = -1;
#pragma unroll
          for (int j = 0; j < C::G; ++j) k = (oi == j) ? ks[j] : k;
          unsigned off = kOOB;
          if (k >= 0) off = (unsigned)((((int64_t)k * ncp + pass * NC) * NB) * 2048) + (unsigned)wi * 16u;
          wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
        }
      };
      auto wstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::WRN; ++i) {
          const int e = tid + i * 256;
          if (e < C::WSLAB / 16) *reinterpret_cast<u32x4 *>(smem + C::O_W + buf * C::WSLAB + e * 16) = wreg[i];
        }
      };
      wload();
      wstore(0);
      LGS_HALO_VMCNT0();                                            // this wave's row pieces have landed
      __syncthreads();                                              // rows, table, klist, slab 0 visible
      if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[0] += tt1 - tt0; tt0 = tt1; }

      const int nslab = (nk + C::G - 1) / C::G;
      int buf = 0;
      for (int slab = 0; slab < nslab; ++slab) {
        const bool more = slab + 1 < nslab;
        if (more) wload();                                          // in flight under this slab's multiply
        if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[1] += tt1 - tt0; tt0 = tt1; tacc[5] += 1; }
        const int kcount = min(C::G, nk - slab * C::G);
        for (int oi = 0; oi < kcount; ++oi) {
          const int k = __builtin_ctz(crem);
          crem &= crem - 1;
          const unsigned s0 = l_tab[k * kHaloT + row0], s1 = l_tab[k * kHaloT + row1];
          const unsigned q0 = s0 - (unsigned)sb, q1 = s1 - (unsigned)sb;     // 0xffff - sb stays >= cnt
          const bool ok0 = q0 < (unsigned)cnt, ok1 = q1 < (unsigned)cnt;
          const bool act0 = __ballot(ok0) != 0ull, act1 = __ballot(ok1) != 0ull;
          if (!act0 && !act1) continue;
          if (tr) tacc[6] += (act0 ? 1 : 0) + (act1 ? 1 : 0);
          const char *a0 = l_rows + (ok0 ? q0 : (unsigned)C::CAP) * C::PITCH + h * 32;
          const char *a1 = l_rows + (ok1 ? q1 : (unsigned)C::CAP) * C::PITCH + h * 32;
          const char *wb = smem + C::O_W + buf * C::WSLAB + oi * C::WK + lane * 16;
          auto body = [&](auto A0, auto A1) __attribute__((always_inline)) {
            constexpr bool B0 = decltype(A0)::value, B1 = decltype(A1)::value;
            // software pipeline over the 2 NC k-steps (16 channels each): the LDS reads of step s + 1 are issued in front of the
            // MFMAs of step s (one wave per SIMD: nobody else covers the LDS latency), interleaved one read per MFMA; the
            // sched_group_barriers pin that order (left alone, hipcc issued read -> lgkmcnt(0) -> 1-3 MFMAs, every latency exposed)
            constexpr int NS = 2 * NC, NR = NB + (B0 ? 1 : 0) + (B1 ? 1 : 0), NM = NB * ((B0 ? 1 : 0) + (B1 ? 1 : 0));
            u32x4 wf[2][NB], f0[2], f1[2];
            auto rd = [&](int st, int b) __attribute__((always_inline)) {
              const int c = st >> 1, t = st & 1;
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) wf[b][nb] = *reinterpret_cast<const u32x4 *>(wb + ((c * NB + nb) * 2 + t) * 1024);
              if constexpr (B0) f0[b] = *reinterpret_cast<const u32x4 *>(a0 + 64 * c + 16 * t);
              if constexpr (B1) f1[b] = *reinterpret_cast<const u32x4 *>(a1 + 64 * c + 16 * t);
            };
            rd(0, 0);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
              const int b = st & 1;
              if (st + 1 < NS) rd(st + 1, b ^ 1);
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                if constexpr (B0)
                  acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[b][nb]), __builtin_bit_cast(bf16x8, f0[b]), acc[0][nb], 0, 0, 0);
                if constexpr (B1)
                  acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[b][nb]), __builtin_bit_cast(bf16x8, f1[b]), acc[1][nb], 0, 0, 0);
              }
            }
            // schedule: [reads of step 0] then per step: MFMA / read alternating, the remaining MFMAs at the end
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
              if (st + 1 < NS) {
#pragma unroll
                for (int i = 0; i < (NR < NM ? NR : NM); ++i) {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
                if (NR > NM) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
              } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
              }
            }
          };
          if (act0 && act1) body(std::true_type{}, std::true_type{});
          else if (act0) body(std::true_type{}, std::false_type{});
          else body(std::false_type{}, std::true_type{});
        }
        if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[2] += tt1 - tt0; tt0 = tt1; }
        if (more) {
          wstore(buf ^ 1);                                          // last read before the previous slab's barrier
          __syncthreads();
          buf ^= 1;
        }
        if (tr) { tt1 = __builtin_amdgcn_s_memtime(); tacc[3] += tt1 - tt0; tt0 = tt1; }
      }
    }
  }

  // ---- epilogue: lane (voxel vx, half h) owns channels nb*32 + 8q + 4h + {0..3} of its two rows
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int64_t p = pos0 + (rb == 0 ? row0 : row1);
    const int32_t orow = v.out_row[p];
    if (orow < 0) continue;
    bf16_t *dst = out + (int64_t)orow * o_real;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = nb * 32 + 8 * q + 4 * h;
        if (c0 >= o_real) continue;
        float o0 = acc[rb][nb][4 * q + 0], o1 = acc[rb][nb][4 * q + 1], o2 = acc[rb][nb][4 * q + 2], o3 = acc[rb][nb][4 * q + 3];
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
  if (tr) {
    tt1 = __builtin_amdgcn_s_memtime(); tacc[4] += tt1 - tt0;
    if (lane == 0)
      for (int i = 0; i < 8; ++i) trace[i] = tacc[i];
  }
}

// ------------------------------------------------------------------------------------------------ host side
namespace {
struct HaloShape { int nc, npass, nb; };
inline bool halo_shape(int g_real, int o_real, HaloShape *hs) {
  if (g_real % 8 != 0 || o_real % 4 != 0 || g_real < 32 || o_real < 8 || o_real > 128) return false;
  const int gc = pad32(g_real) / 32, nb = pad32(o_real) / 32;
  int nc, npass;
  if (gc <= 3) { nc = gc; npass = 1; }
  else if (gc == 4) { nc = 2; npass = 2; }
  else if (gc == 6) { nc = 3; npass = 2; }
  else return false;
  // instantiated (NC, NB) pairs: the shapes of Res16UNet34C / 14A / 18 at levels 0-2 in both directions
  const bool have = (nc == 1 && (nb == 1 || nb == 2)) || (nc == 2 && nb >= 1 && nb <= 4) || (nc == 3 && (nb == 3 || nb == 4));
  if (!have) return false;
  hs->nc = nc; hs->npass = npass; hs->nb = nb;
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
  *ncp = hs.nc * hs.npass; *nbp = hs.nb;
  return (int64_t)27 * (*ncp) * (*nbp) * 2 * 64;     // 16-byte units
}

template <int NC, int NB>
static int launch_halo_t(const HaloView &hv, const void *in, int g_real, int in_ld, int npass, const void *wp, int ncp, void *out, int o_real,
                         const float *bias, int accum, unsigned in_bytes, unsigned w_bytes, hipStream_t s) {
  using C = HaloCfg<NC, NB>;
  static bool attr_set = false;
  if (!attr_set) {
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_halo<NC, NB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_halo<NC, NB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
    attr_set = true;
  }
  const unsigned nt = (unsigned)(hv.v.n_pad / kHaloT);
  if (tune(T_HALO_TRACE) != 0) {
    static unsigned long long *trace = nullptr;
    if (!trace) LGS_HIP(hipMalloc(&trace, 8 * sizeof(unsigned long long)));
    LGS_HIP(hipMemsetAsync(trace, 0, 8 * sizeof(unsigned long long), s));
    LGS_KLAUNCH((k_conv_halo<NC, NB, true>), dim3(nt), dim3(256), C::LDS, s, hv, reinterpret_cast<const bf16_t *>(in), g_real, in_ld, npass,
                reinterpret_cast<const u32x4 *>(wp), ncp, reinterpret_cast<bf16_t *>(out), o_real, bias, accum, in_bytes, w_bytes, trace);
    unsigned long long h[8];
    LGS_HIP(hipStreamSynchronize(s));
    LGS_HIP(hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[k_conv_halo trace] %d->%d NC %d NB %d: tile of %llu distinct rows, %llu slabs, %llu active (block, offset) pairs in wave 0; cycles: "
            "prologue %llu  weight issue %llu  multiply %llu  slab end (store + barrier) %llu  epilogue %llu\n", g_real, o_real, NC, NB, h[7], h[5], h[6],
            h[0], h[1], h[2], h[3], h[4]);
    return 0;
  }
  LGS_KLAUNCH((k_conv_halo<NC, NB, false>), dim3(nt), dim3(256), C::LDS, s, hv, reinterpret_cast<const bf16_t *>(in), g_real, in_ld, npass,
              reinterpret_cast<const u32x4 *>(wp), ncp, reinterpret_cast<bf16_t *>(out), o_real, bias, accum, in_bytes, w_bytes,
              (unsigned long long *)nullptr);
  LGS_HIP(hipGetLastError());
  return 0;
}

int launch_conv_halo(const HaloView &hv, int mirror, const void *in, int g_real, int in_ld, const void *wp, int ncp, int nbp,
                     void *out, int o_real, const float *bias, int accum, hipStream_t s) {
  (void)mirror;                                      // the dgrad mirroring (K - 1 - k) is folded into the weight packing
  HaloShape hs;
  LGS_REQUIRE(halo_shape(g_real, o_real, &hs) && ncp == hs.nc * hs.npass && nbp == hs.nb, "halo conv: shape / packed-image mismatch (internal error)");
  const int ld = in_ld > 0 ? in_ld : g_real;
  const uint64_t in_bytes64 = (uint64_t)hv.v.n_in * (uint64_t)ld * 2, w_bytes64 = (uint64_t)27 * ncp * nbp * 2048;
  LGS_REQUIRE(in_bytes64 < 0xfffff000ull && w_bytes64 < 0xfffff000ull && (ld * 2) % 16 == 0,
              "sparse conv: a feature or weight tensor of 4 GiB or more is beyond the 32-bit buffer-descriptor path");
#define LGS_HALO_CASE(NCV, NBV)                                                                                              \
  if (hs.nc == NCV && hs.nb == NBV)                                                                                          \
    return launch_halo_t<NCV, NBV>(hv, in, g_real, ld, hs.npass, wp, ncp, out, o_real, bias, accum, (unsigned)in_bytes64, (unsigned)w_bytes64, s);
  LGS_HALO_CASE(1, 1) LGS_HALO_CASE(1, 2)
  LGS_HALO_CASE(2, 1) LGS_HALO_CASE(2, 2) LGS_HALO_CASE(2, 3) LGS_HALO_CASE(2, 4)
  LGS_HALO_CASE(3, 3) LGS_HALO_CASE(3, 4)
#undef LGS_HALO_CASE
  LGS_REQUIRE(false, "halo conv: no kernel instance for this shape (internal error)");
}

}  // namespace lgs
