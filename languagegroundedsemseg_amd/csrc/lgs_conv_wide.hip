// lgs_conv_wide.hip -- sparse convolution forward / dgrad for WIDE channel counts (>= 256 output channels, bf16) on gfx950.
//
// Serves the same call sites as lgs_conv.hip (MinkowskiConvolution forward + autograd dgrad,
//   /root/reference/models/modules/common.py:195-203, models/modules/resnet_block.py:41-57)
// for the launch shapes of the CLIP representation model Res16UNet34D (/root/reference/models/clip_models.py:205-215,
// scripts/text_representation_train.sh:7): 3^3 convolutions 512 -> 512, 544 -> 512, 256 -> 256 ... on the big maps.
//
// Why a second kernel.  At these widths the convolution is COMPUTE bound (7.2 TFLOP per level-0 512 -> 512 launch against
// ~3 GB of compulsory HBM traffic), and k_conv_gather -- built for the HBM-bound narrow layers: every wave gathers its own
// 32 rows straight into MFMA operand registers and reads ALL weight fragments of the tile from LDS -- sits on the CU's
// vector-memory instruction rate and on a per-slab barrier that every wave reaches at the pace of its own gathers
// (round 2: 586 TFLOP/s stand-alone, 52 % of the wave time in s_barrier, MFMA pipe 31 % busy).
//
// This kernel is a 2-D blocked implicit GEMM in the shape of a dense GEMM main loop:
//   * workgroup tile = 256 positions x 256 output channels, 8 waves as 2 (positions) x 4 (channels); a wave owns
//     128 positions x 64 channels = 4 x 2 MFMA tiles (32x32x16 bf16), 128 accumulator registers;
//   * one STAGE = (kernel offset k, 64 input channels): the 256 gathered row pieces (128 B each, 32 KB) and the offset's
//     64 x 256 weight block (32 KB, pre-packed in fragment order) are brought into LDS by LDS-DMA (buffer_load ... lds), 8
//     instructions per wave, shared by all eight waves -- each gathered byte and each weight byte is fetched once per
//     tile and read from LDS 4 x / 2 x; a missing neighbour is an out-of-range offset (zeros, no traffic);
//   * two LDS stage buffers: the DMA of stage s+1 is in flight while stage s is multiplied; ONE barrier per stage
//     (32 MFMAs per wave), waits are explicit counted s_waitcnt, LDS reads of the loop are inline asm (hipcc would fence
//     every ds_read behind a pending LDS-DMA with vmcnt(0));
//   * the gathered tile is row-major [256][128 B]; a B fragment read (32 rows x 16 B) would be an 8-way bank conflict,
//     so the 16-byte pieces of a row are XOR-swizzled with (row >> 1) & 7 -- applied to the per-lane SOURCE address of
//     the DMA (its LDS side is lane-linear) and to the read address;
//   * 32-row blocks without a single neighbour at an offset skip their LDS reads and MFMAs (rows are sorted by
//     neighbourhood shape inside Morton windows, so such blocks are common: padded MFMA work 1.3 x the real pairs); the two
//     position halves of the wave grid take alternating 32-row blocks so both see the same amount of work;
//   * reduction order: groups of 128 input channels outermost, then the offsets, then the two 64-channel stages of the
//     group -- a gathered row's 256-byte segment serves all offsets back to back (L2 working set, as in k_conv_gather).
// Output rows are written exactly once (no atomics, deterministic).  dgrad is the same kernel on the transposed /
// mirrored packed weights, exactly as for k_conv_gather (the packed image layout is shared: slabs of 2 chunks, 8 blocks).
#include "lgs_common.h"

#include <stdlib.h>

namespace lgs {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
#define LGS_AS3(p) ((__attribute__((address_space(3))) void *)(p))
#define LGS_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14)))
// The 8 accumulator tiles live in FIXED accumulation registers a[0:127] (the unified register file is split 128 + 128 at
// two waves per SIMD): tile (row block j, column block n) = a[32 j + 16 n : 32 j + 16 n + 15].  They are NOT C++ values:
// the asm statements of the main loop and of the epilogue name them literally and list all of a0..a127 as clobbered, so
// the compiler never places anything of its own there (it has no matrix builtins to allocate in this kernel, and the
// VGPR side stays far below its 128: nothing is spilled to accumulation registers either -- the build checks the ISA).
// Left to the register allocator (builtin MFMAs, "+a" operands, even physical-register constraints on C++ values) hipcc
// shuffled accumulator tuples through VGPRs and scratch inside the main loop.
#define LGS_ACC_CLOBBER "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define LGS_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off) : "memory")

constexpr int kWideTM = 256;                       // positions per workgroup tile
constexpr int kWideA = 32768, kWideW = 32768;      // bytes per stage buffer: gathered tile, weight block
constexpr int kWideAOff = 0, kWideWOff = 2 * kWideA;                  // [A0][A1][W0][W1]
constexpr int kWideIdxOff = 2 * kWideA + 2 * kWideW;                  // int32 [27][256] gather rows of the tile
constexpr int kWideActOff = kWideIdxOff + 27 * kWideTM * 4;           // uint8 [27][8]: 32-row block has a neighbour at the offset
constexpr int kWideLds = kWideActOff + 27 * 8 + 40;

// iterator over the stages of a tile: channel groups of gc64 64-channel stages outermost, then the offsets of `smask`
struct WideIter {
  uint32_t smask, rem;
  int nc64, gc64, gbase, gend, c, slot;
  __device__ __forceinline__ void init(uint32_t sm, int nc, int gc) {
    smask = sm; rem = sm; nc64 = nc; gc64 = gc; gbase = 0; gend = min(gc, nc); c = gend; slot = -1;
  }
  __device__ __forceinline__ bool next() {
    if (++c < gend) return true;
    if (rem == 0) {
      gbase = gend;
      if (gbase >= nc64) return false;
      gend = min(gbase + gc64, nc64);
      rem = smask;
    }
    if (rem == 0) return false;
    slot = __builtin_ctz(rem);
    rem &= rem - 1;
    c = gbase;
    return true;
  }
};

__global__ __launch_bounds__(512, 2) void k_conv_wide(View v, const bf16_t *__restrict__ in, int cin_real, int nc64,
                                                       const u32x4 *__restrict__ wp, int nb_total, int ncp, int nbp,
                                                       bf16_t *__restrict__ out, int cout_real, const float *__restrict__ bias,
                                                       unsigned in_bytes, unsigned w_bytes, int in_ld, int gc64, int ny, int sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int vx = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // XCD-aware order (speed only): the dispatcher places workgroup b on XCD b % 8.  Every XCD gets a contiguous run of position
  // tiles (neighbouring tiles gather the same rows) and -- when the output has 2 / 4 / 8 channel tiles -- ONE channel tile:
  // XCD x only ever streams the weights of channel tile x % ny, so the weights of the channel group in flight (27 offsets x
  // 64 channels x 256 outputs = 0.9 MB) stay in its 4 MB L2 while its 32 workgroups walk through them (with both channel
  // tiles on one XCD the L2 hit rate was 50 % and the weight stream came from the Infinity Cache)
  int64_t tile;
  int ytile;
  {
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    if ((8 % ny) == 0 && (nwg % 8u) == 0) {
      const unsigned ntile = (unsigned)(v.n_pad / kWideTM), g = 8u / (unsigned)ny;      // g XCDs share a channel tile
      const unsigned sub = xcd / (unsigned)ny, per = (ntile + g - 1) / g;    // XCD's slice of the position tiles
      ytile = (int)(xcd % (unsigned)ny);
      tile = (int64_t)sub * per + j;
      if (j >= per || tile >= (int64_t)ntile) return;                        // (grid padded to a multiple of 8 per slice)
    } else {
      const unsigned q = nwg >> 3, r = nwg & 7u;
      const unsigned idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
      ytile = (int)(idx % (unsigned)ny);
      tile = idx / (unsigned)ny;
    }
  }
  const int64_t pos_wg = tile * kWideTM;
  const int nb_wg = ytile * 8;

  // ---- offsets this tile visits (64-position ballot masks of the map), parked gather rows
  uint32_t smask = 1;
  int kw_single = 0;
  if (v.KS > 1) {
    smask = 0;
#pragma unroll
    for (int g = 0; g < kWideTM / 64; ++g) smask |= v.mask64[pos_wg / 64 + g];
  } else if (v.tile_k) {
    kw_single = v.tile_k[pos_wg / 64];
    if (kw_single < 0) return;            // padding group of a grouped (strided / transposed) view
  }
  smask = __builtin_amdgcn_readfirstlane(smask);
  kw_single = __builtin_amdgcn_readfirstlane(kw_single);
  // the kernel-map rows are kept in LDS as BYTE OFFSETS of the gathered rows (row * row stride; 0xfffff000 = no neighbour: any
  // offset built on it is out of the buffer's range and reads zeros), computed once per tile instead of once per stage
  constexpr unsigned kNoRow = 0xfffff000u;
  const unsigned row_bytes_p = (unsigned)in_ld * 2u;
  uint32_t *l_idx = reinterpret_cast<uint32_t *>(smem + kWideIdxOff);
  unsigned char *l_act = reinterpret_cast<unsigned char *>(smem + kWideActOff);
  {
    const int r = tid & (kWideTM - 1), par = tid >> 8;        // threads 0..255: even offsets, 256..511: odd offsets
    if (v.nbr) {
      int32_t tmp[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        const int sl = 2 * i + par;
        if (sl < 27 && ((smask >> sl) & 1u)) tmp[i] = v.nbr[(int64_t)sl * v.n_pad + pos_wg + r];
      }
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        const int sl = 2 * i + par;
        if (sl < 27 && ((smask >> sl) & 1u)) l_idx[sl * kWideTM + r] = tmp[i] >= 0 ? (uint32_t)tmp[i] * row_bytes_p : kNoRow;
      }
    } else if (par == 0) {
      const int64_t p = pos_wg + r;
      l_idx[r] = p < v.n_in ? (uint32_t)p * row_bytes_p : kNoRow;
    }
  }
  __syncthreads();
  // which 32-row blocks have a neighbour at which offset (exact: the map's masks are per 64 positions)
#pragma unroll 1
  for (int s2 = 0; s2 < 28; s2 += 2) {
    const int sl = s2 + h;
    const bool ok = sl < 27 && ((smask >> sl) & 1u) && l_idx[sl * kWideTM + wave * 32 + vx] != kNoRow;
    const uint64_t b = __ballot(ok);
    if (lane == 0) {
      l_act[s2 * 8 + wave] = (b & 0xffffffffull) != 0ull;
      if (s2 + 1 < 27) l_act[(s2 + 1) * 8 + wave] = (b >> 32) != 0ull;
    }
  }
  __syncthreads();
  // Per-offset facts of this wave sit in lane `offset` of two VGPRs and are fetched with ONE v_readlane per stage (the scalar
  // unit is shared by the CU's eight waves: ~80 scalar instructions per wave and stage cost as much as the MFMAs):
  //   stab bits 0..3: this wave's row block j (= tile block 2 j + wm) has a neighbour; bits 4,5: the two 32-row blocks this wave
  //   GATHERS (wm = 1: tile blocks 2 (wave - 4), 2 (wave - 4) + 1) have one;   wtab: byte offset of the offset's weight block
  uint32_t stab = 0, wtab = 0;
  {
    bool nz = false;
    if (lane < 27) {
      const unsigned char *ab = l_act + lane * 8;
      nz = *reinterpret_cast<const uint64_t *>(ab) != 0ull;
      stab = (ab[wm] != 0 ? 1u : 0u) | (ab[2 + wm] != 0 ? 2u : 0u) | (ab[4 + wm] != 0 ? 4u : 0u) | (ab[6 + wm] != 0 ? 8u : 0u);
      const int gb = 2 * (wave & 3);
      stab |= (ab[gb] != 0 ? 16u : 0u) | (ab[gb + 1] != 0 ? 32u : 0u);
      const int kw = v.KS > 1 ? lane : kw_single;
      wtab = (unsigned)((((int64_t)kw * ncp) * nbp + nb_wg) * (2 * 64) * 16);
    }
    smask = (uint32_t)(__ballot(nz) & 0x7ffffffull);      // exact set of offsets with at least one pair in this tile
  }

  // accumulators: a[0:127], see LGS_ACC_CLOBBER
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" ::: LGS_ACC_CLOBBER);
  constexpr unsigned kOOB = 0xfffff000u;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wp), 0, (int)w_bytes, 0x00020000);

  // ---- DMA side.  ROLES: the two waves of a SIMD (w, w + 4: wm = 0 / 1) run complementary schedules, so that one of them
  // multiplies while the other sits in the vector-memory issue queue (every wave issuing its DMA right behind the barrier and
  // multiplying afterwards made transfer and multiply add up: knock-out runs DMA only 6.2 ms, multiply only 7.3 ms, both
  // 10.4 ms; a 1 KB DMA instruction occupies the CU's address unit for ~16 cycles, 64 of them per stage):
  //   wm = 0 (waves 0..3): the WEIGHT block of stage s+1, 8 instructions of 1 KB, at the start of stage s, then the multiply;
  //   wm = 1 (waves 4..7): row blocks 0, 1 of the multiply first, then the GATHERS of stage s+1 for tile blocks 2 w', 2 w' + 1
  //                         (8 instructions of 8 rows x 128 B; none for a block without a neighbour), then row blocks 2, 3.
  const int g_r = lane >> 3;                                   // row inside the 8-row instruction
  const int g_p = lane & 7;                                    // 16-byte piece of the 128-byte row segment (LDS side)
  const int wq = wave & 3;
  const unsigned idx_rd = lds0 + (unsigned)kWideIdxOff + (unsigned)((wq * 64 + g_r) * 4);
  // source piece (swizzled with (row >> 1) & 7, row = 64 wq + 8 jj + g_r: depends on the parity of jj only), bytes
  const unsigned g_src0 = (unsigned)((g_p ^ ((g_r >> 1) & 7)) * 16), g_src1 = (unsigned)((g_p ^ ((4 + (g_r >> 1)) & 7)) * 16);
  const unsigned w_voff = (unsigned)(lane * 16), w_voff4 = (unsigned)(lane * 16 + 4096);
  auto issue_w = [&](uint32_t wbase, int c64, int buf) __attribute__((always_inline)) {
    // packed image: [kw][chunk32][block][t][lane] x 16 B; a stage = chunks 2 c64, 2 c64 + 1, blocks nb_wg .. nb_wg + 7: two
    // contiguous 16 KB pieces; wave wq copies KB 8 wq .. 8 wq + 7 of the 32 (chunk wq >> 1, second half for odd wq)
    char *wdst = smem + kWideWOff + buf * kWideW + wq * 8192;
    const unsigned sbase = wbase + (unsigned)((c64 * 2 + (wq >> 1)) * nbp) * 2048u + (unsigned)((wq & 1) * 8192);
    // the instruction's immediate offset is added to the memory address AND to the LDS address (M0 base + offset + 16 lane):
    // one M0 value serves four instructions
#define LGS_WIDE_WDMA(BASE, VOFF, IMM) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LGS_AS3(wdst + (BASE)), 16, VOFF, sbase, IMM, 0)
    LGS_WIDE_WDMA(0, w_voff, 0); LGS_WIDE_WDMA(0, w_voff, 1024); LGS_WIDE_WDMA(0, w_voff, 2048); LGS_WIDE_WDMA(0, w_voff, 3072);
    LGS_WIDE_WDMA(4096, w_voff4, 0); LGS_WIDE_WDMA(4096, w_voff4, 1024); LGS_WIDE_WDMA(4096, w_voff4, 2048); LGS_WIDE_WDMA(4096, w_voff4, 3072);
#undef LGS_WIDE_WDMA
  };
  auto issue_a = [&](int slot, uint32_t st, int c64, int buf) __attribute__((always_inline)) {
    char *adst = smem + kWideAOff + buf * kWideA + wq * 8192;
    const unsigned cb = (unsigned)(c64 * 128), lim = (unsigned)cin_real * 2u;
    // pieces beyond the row's channels (the last 64-channel stage of e.g. 544 channels) must read zeros, not the next row
    const unsigned c0 = cb + g_src0, c1 = cb + g_src1;
    const bool ok0 = c0 + 16u <= lim, ok1 = c1 + 16u <= lim;
    const unsigned ird = idx_rd + (unsigned)(slot * kWideTM * 4);
#define LGS_WIDE_GATHER4(JJ0)                                                                                               \
    {                                                                                                                       \
      uint32_t r0, r1, r2, r3;                                                                                              \
      asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\t"      \
                   "ds_read_b32 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"                                                   \
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                                             \
                   : "v"(ird), "n"((JJ0) * 32), "n"((JJ0) * 32 + 32), "n"((JJ0) * 32 + 64), "n"((JJ0) * 32 + 96) : "memory"); \
      const unsigned o0 = ok0 ? r0 + c0 : kOOB, o1 = ok1 ? r1 + c1 : kOOB, o2 = ok0 ? r2 + c0 : kOOB, o3 = ok1 ? r3 + c1 : kOOB;   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, LGS_AS3(adst + (JJ0) * 1024), 16, o0, 0, 0, 0);                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, LGS_AS3(adst + (JJ0) * 1024 + 1024), 16, o1, 0, 0, 0);                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, LGS_AS3(adst + (JJ0) * 1024 + 2048), 16, o2, 0, 0, 0);                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, LGS_AS3(adst + (JJ0) * 1024 + 3072), 16, o3, 0, 0, 0);                \
    }
    if (st & 16u) LGS_WIDE_GATHER4(0)
    if (st & 32u) LGS_WIDE_GATHER4(4)
#undef LGS_WIDE_GATHER4
  };

  // ---- compute side
  // gathered fragment of (row block j, step s16): row (2 j + wm) 32 + vx, piece q = (c32 << 2 | h << 1 | t) ^ ((row >> 1) & 7)
  // [(row >> 1) & 7 == (vx >> 1) & 7: the block offset is a multiple of 32]
  unsigned a_lane[4];
  {
    const unsigned L = (unsigned)(((vx >> 1) & 7) ^ (h << 1));
#pragma unroll
    for (int s16 = 0; s16 < 4; ++s16) {
      const unsigned S = (unsigned)(((s16 >> 1) << 2) | (s16 & 1));
      a_lane[s16] = lds0 + (unsigned)kWideAOff + (unsigned)((wm * 32 + vx) * 128) + ((L ^ S) << 4);
    }
  }
  const unsigned w_lane = lds0 + (unsigned)kWideWOff + (unsigned)(wn * 4096 + lane * 16);
  // One stage of one wave = one asm statement per 32-row block (the DMA of the next stage goes in between).  The first one
  // reads the 8 weight fragments (4 k-steps x 2 column blocks, kept in w0..w7 for the other three); every ACTIVE block
  // (bit j of m4) reads its 4 gathered fragments and runs its 8 MFMAs on its fixed accumulators, a block without a
  // neighbour at this offset branches over both.  Control flow stays inside the asm: any compiler-visible branch around
  // accumulator updates made hipcc copy 16-register tuples through VGPRs and scratch.  The LDS latency in front of a
  // block's MFMAs is covered by the other wave of the SIMD.     fragment w{2k+n} = k-step k, column block n
  u32x4 w0, w1, w2, w3, w4, w5, w6, w7;
  unsigned wb, ab0, ab1, ab2, ab3;
#define LGS_WIDE_MFMAS(C0, C1)                                                     \
                 "v_mfma_f32_32x32x16_bf16 " C0 ", %[w0], %[a0], " C0 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C1 ", %[w1], %[a0], " C1 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C0 ", %[w2], %[a1], " C0 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C1 ", %[w3], %[a1], " C1 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C0 ", %[w4], %[a2], " C0 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C1 ", %[w5], %[a2], " C1 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C0 ", %[w6], %[a3], " C0 "\n\t"        \
                 "v_mfma_f32_32x32x16_bf16 " C1 ", %[w7], %[a3], " C1 "\n"
#define LGS_WIDE_AREADS(OFF)                                                       \
                 "ds_read_b128 %[a0], %[p0] offset:" #OFF "\n\t"                   \
                 "ds_read_b128 %[a1], %[p1] offset:" #OFF "\n\t"                   \
                 "ds_read_b128 %[a2], %[p2] offset:" #OFF "\n\t"                   \
                 "ds_read_b128 %[a3], %[p3] offset:" #OFF "\n\t"                   \
                 "s_waitcnt lgkmcnt(0)\n\t"
  auto block0 = [&](uint32_t m4, int buf) __attribute__((always_inline)) {
    wb = w_lane + (unsigned)(buf * kWideW);
    ab0 = a_lane[0] + (unsigned)(buf * kWideA); ab1 = a_lane[1] + (unsigned)(buf * kWideA);
    ab2 = a_lane[2] + (unsigned)(buf * kWideA); ab3 = a_lane[3] + (unsigned)(buf * kWideA);
    u32x4 a0, a1, a2, a3;
    asm volatile("s_cmp_eq_u32 %[m], 0\n\t"
                 "s_cbranch_scc1 .Lwide_b0_%=\n\t"
                 "ds_read_b128 %[w0], %[wb] offset:0\n\t"
                 "ds_read_b128 %[w1], %[wb] offset:2048\n\t"
                 "ds_read_b128 %[w2], %[wb] offset:1024\n\t"
                 "ds_read_b128 %[w3], %[wb] offset:3072\n\t"
                 "ds_read_b128 %[w4], %[wb] offset:16384\n\t"
                 "ds_read_b128 %[w5], %[wb] offset:18432\n\t"
                 "ds_read_b128 %[w6], %[wb] offset:17408\n\t"
                 "ds_read_b128 %[w7], %[wb] offset:19456\n\t"
                 "s_bitcmp0_b32 %[m], 0\n\t"
                 "s_cbranch_scc1 .Lwide_b0w_%=\n\t"
                 LGS_WIDE_AREADS(0)
                 LGS_WIDE_MFMAS("a[0:15]", "a[16:31]")
                 "s_branch .Lwide_b0_%=\n"
                 ".Lwide_b0w_%=:\n\t"
                 "s_waitcnt lgkmcnt(0)\n"      /* w0..w7 are complete when the statement ends: the compiler may move them */
                 ".Lwide_b0_%=:"
                 : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [w5] "=&v"(w5), [w6] "=&v"(w6),
                   [w7] "=&v"(w7), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
                 : [m] "s"(m4), [wb] "v"(wb), [p0] "v"(ab0), [p1] "v"(ab1), [p2] "v"(ab2), [p3] "v"(ab3)
                 : "memory", "scc", LGS_ACC_CLOBBER);
  };
#define LGS_WIDE_BLOCKN(NAME, J, OFF, C0, C1)                                                                              \
  auto NAME = [&](uint32_t m4) __attribute__((always_inline)) {                                                           \
    u32x4 a0, a1, a2, a3;                                                                                                  \
    asm volatile("s_bitcmp0_b32 %[m], " #J "\n\t"                                                                          \
                 "s_cbranch_scc1 .Lwide_bn_%=\n\t"                                                                         \
                 LGS_WIDE_AREADS(OFF)                                                                                      \
                 LGS_WIDE_MFMAS(C0, C1)                                                                                    \
                 ".Lwide_bn_%=:"                                                                                           \
                 : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)                                          \
                 : [m] "s"(m4), [p0] "v"(ab0), [p1] "v"(ab1), [p2] "v"(ab2), [p3] "v"(ab3), [w0] "v"(w0), [w1] "v"(w1),      \
                   [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4), [w5] "v"(w5), [w6] "v"(w6), [w7] "v"(w7)                      \
                 : "memory", "scc", LGS_ACC_CLOBBER);                                                                      \
  };
  LGS_WIDE_BLOCKN(block1, 1, 8192, "a[32:47]", "a[48:63]")
  LGS_WIDE_BLOCKN(block2, 2, 16384, "a[64:79]", "a[80:95]")
  LGS_WIDE_BLOCKN(block3, 3, 24576, "a[96:111]", "a[112:127]")
#undef LGS_WIDE_BLOCKN

  // ---- main loop: [wait own DMA of stage s] [barrier] then the multiply of stage s with this wave's share of the DMA of
  // stage s+1 (other buffer).  cur_* = stage being multiplied, nxt_* = stage being fetched (one iterator, one step ahead).
  WideIter it;
  it.init(smask, nc64, gc64);
  bool have = it.next();
  int cur_slot = it.slot;
  if (have) {                                                      // prologue: stage 0 into buffer 0
    if (wm == 0) issue_w(__builtin_amdgcn_readlane(wtab, it.slot), it.c, 0);
    else issue_a(it.slot, __builtin_amdgcn_readlane(stab, it.slot), it.c, 0);
  }
  int buf = 0;
  // (per-phase shader-clock sums and the knock-out bits of the experiment builds: tools/dbg/conv_wide_instrumentation.patch)
  while (have) {
    const bool more = it.next();                                   // `it` now names stage s+1
    LGS_VMCNT(0);                                                  // this wave's pieces of stage s have landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // everybody's have; buffer buf ^ 1 is no longer read
    const uint32_t m4 = __builtin_amdgcn_readlane(stab, cur_slot) & 15u;
    // where the DMA of stage s+1 goes between the four row blocks of stage s: tuning knob WIDE_SCHED (kernel-uniform branches)
    if (wm == 0) {
      const int at = sched == 0 || sched == 4 ? 0 : sched == 2 ? 1 : sched == 3 ? 3 : 2;     // weights before block `at`
      if (more && at == 0) issue_w(__builtin_amdgcn_readlane(wtab, it.slot), it.c, buf ^ 1);
      block0(m4, buf);
      if (more && at == 1) issue_w(__builtin_amdgcn_readlane(wtab, it.slot), it.c, buf ^ 1);
      block1(m4);
      if (more && at == 2) issue_w(__builtin_amdgcn_readlane(wtab, it.slot), it.c, buf ^ 1);
      block2(m4);
      if (more && at == 3) issue_w(__builtin_amdgcn_readlane(wtab, it.slot), it.c, buf ^ 1);
      block3(m4);
    } else {
      if (more && sched != 0) issue_a(it.slot, __builtin_amdgcn_readlane(stab, it.slot), it.c, buf ^ 1);
      block0(m4, buf);
      block1(m4);
      if (more && sched == 0) issue_a(it.slot, __builtin_amdgcn_readlane(stab, it.slot), it.c, buf ^ 1);
      block2(m4);
      block3(m4);
    }
    cur_slot = it.slot;
    have = more;
    buf ^= 1;
  }
  LGS_VMCNT(0);
  // the MFMAs are opaque asm: the compiler does not know that the accumulators were just written by the matrix pipe
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: lane (voxel vx, half h) owns channels nb 32 + 8 q + 4 h + {0..3} of its rows; one accumulator tile at a
  // time is read out of its fixed accumulation registers
  int32_t orow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t p = pos_wg + (2 * j + wm) * 32 + vx;
    orow[j] = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
  }
#define LGS_WIDE_STORE(J, N, READS)                                                                                       \
  {                                                                                                                       \
    float t[16];                                                                                                          \
    asm volatile(READS                                                                                                    \
                 : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]), "=v"(t[4]), "=v"(t[5]), "=v"(t[6]), "=v"(t[7]), "=v"(t[8]), \
                   "=v"(t[9]), "=v"(t[10]), "=v"(t[11]), "=v"(t[12]), "=v"(t[13]), "=v"(t[14]), "=v"(t[15])                    \
                 : : "memory");                                                                                           \
    const int nb = nb_wg + wn * 2 + (N);                                                                                  \
    if (orow[J] >= 0 && nb < nb_total) {                                                                                  \
      bf16_t *dst = out + (int64_t)orow[J] * cout_real;                                                                   \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                     \
        const int c0 = nb * 32 + 8 * q + 4 * h;                                                                           \
        if (c0 < cout_real) {                                                                                             \
          float o0 = t[4 * q + 0], o1 = t[4 * q + 1], o2 = t[4 * q + 2], o3 = t[4 * q + 3];                               \
          if (bias) { o0 += bias[c0]; o1 += bias[c0 + 1]; o2 += bias[c0 + 2]; o3 += bias[c0 + 3]; }                       \
          uint2 pk;                                                                                                       \
          pk.x = (uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16);                                           \
          pk.y = (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16);                                           \
          *reinterpret_cast<uint2 *>(dst + c0) = pk;                                                                      \
        }                                                                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  LGS_WIDE_STORE(0, 0, "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\t" "v_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t" "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\t" "v_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\t" "v_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\t" "v_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\t" "v_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\t" "v_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15")
  LGS_WIDE_STORE(0, 1, "v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\t" "v_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\t" "v_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\t" "v_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\t" "v_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\t" "v_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\t" "v_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\t" "v_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31")
  LGS_WIDE_STORE(1, 0, "v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\t" "v_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\t" "v_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\t" "v_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\t" "v_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\t" "v_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\t" "v_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\t" "v_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47")
  LGS_WIDE_STORE(1, 1, "v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\t" "v_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\t" "v_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\t" "v_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\t" "v_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\t" "v_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\t" "v_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\t" "v_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63")
  LGS_WIDE_STORE(2, 0, "v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\t" "v_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\t" "v_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\t" "v_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\t" "v_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\t" "v_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\t" "v_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\t" "v_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79")
  LGS_WIDE_STORE(2, 1, "v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\t" "v_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\t" "v_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\t" "v_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\t" "v_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\t" "v_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\t" "v_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\t" "v_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95")
  LGS_WIDE_STORE(3, 0, "v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\t" "v_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99\n\t" "v_accvgpr_read_b32 %4, a100\n\tv_accvgpr_read_b32 %5, a101\n\t" "v_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a103\n\t" "v_accvgpr_read_b32 %8, a104\n\tv_accvgpr_read_b32 %9, a105\n\t" "v_accvgpr_read_b32 %10, a106\n\tv_accvgpr_read_b32 %11, a107\n\t" "v_accvgpr_read_b32 %12, a108\n\tv_accvgpr_read_b32 %13, a109\n\t" "v_accvgpr_read_b32 %14, a110\n\tv_accvgpr_read_b32 %15, a111")
  LGS_WIDE_STORE(3, 1, "v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\t" "v_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115\n\t" "v_accvgpr_read_b32 %4, a116\n\tv_accvgpr_read_b32 %5, a117\n\t" "v_accvgpr_read_b32 %6, a118\n\tv_accvgpr_read_b32 %7, a119\n\t" "v_accvgpr_read_b32 %8, a120\n\tv_accvgpr_read_b32 %9, a121\n\t" "v_accvgpr_read_b32 %10, a122\n\tv_accvgpr_read_b32 %11, a123\n\t" "v_accvgpr_read_b32 %12, a124\n\tv_accvgpr_read_b32 %13, a125\n\t" "v_accvgpr_read_b32 %14, a126\n\tv_accvgpr_read_b32 %15, a127")
#undef LGS_WIDE_STORE
}

// bf16 only; the packed weight image has the layout of k_conv_gather's wide tile (slabs of 2 chunks, 8-block tiles)
int launch_conv_wide(const View &v, const void *in, int cin_real, int in_ld, const void *wp, int nb_total, int ncp, int nbp, int K,
                     void *out, int cout_real, const float *bias, int gc64, hipStream_t s) {
  if (v.n_pad == 0) return 0;
  LGS_REQUIRE(v.n_pad % kWideTM == 0 && ncp % 2 == 0 && nbp % 8 == 0 && cin_real % 8 == 0 && cout_real % 4 == 0,
              "wide conv: tile / packed-image layout mismatch (internal error)");
  const int ld = in_ld > 0 ? in_ld : cin_real;
  const uint64_t in_bytes64 = (uint64_t)v.n_in * (uint64_t)ld * 2, w_bytes64 = (uint64_t)K * ncp * nbp * 2 * 64 * 16;
  LGS_REQUIRE(in_bytes64 < 0xfffff000ull && w_bytes64 < 0xfffff000ull,
              "sparse conv: a feature or weight tensor of 4 GiB or more is beyond the 32-bit buffer-descriptor path");
  static bool attr_set = false;
  if (!attr_set) {
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_wide), hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds));
    attr_set = true;
  }
  const int nc64 = (cin_real + 63) / 64, ny = nbp / 8;
  const int gc_env = (int)tune(T_WIDE_GC64);   // tuning knob: 64-channel stages per reduction group
  gc64 = gc_env > 0 ? gc_env : gc64;
  unsigned nwg = (unsigned)(v.n_pad / kWideTM) * (unsigned)ny;
  if (8 % ny == 0) {      // every XCD owns one channel tile and a slice of the position tiles (see the kernel): pad the slices
    const unsigned ntile = (unsigned)(v.n_pad / kWideTM), g = 8u / (unsigned)ny, per = (ntile + g - 1) / g;
    nwg = per * 8u;
  }
  LGS_KLAUNCH(k_conv_wide, dim3(nwg), dim3(512), kWideLds, s, v, reinterpret_cast<const bf16_t *>(in), cin_real, nc64,
              reinterpret_cast<const u32x4 *>(wp), nb_total, ncp, nbp, reinterpret_cast<bf16_t *>(out), cout_real, bias,
              (unsigned)in_bytes64, (unsigned)w_bytes64, ld, gc64 > 0 ? gc64 : nc64, ny, (int)tune(T_WIDE_SCHED));
  LGS_HIP(hipGetLastError());
  return 0;
}

}  // namespace lgs
