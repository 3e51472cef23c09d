// lgs_wgrad_wide.hip -- weight gradient of WIDE 3^3 sparse convolutions (>= 256 input and output channels, bf16) on gfx950.
//
// Serves lgs_conv_wgrad (autograd backward of MinkowskiConvolution, /root/reference/models/modules/common.py:195-203) for the
// launch shapes of the CLIP representation model Res16UNet34D (/root/reference/models/clip_models.py:205-215): 512 -> 512,
// 544 -> 512, 256 -> 256 ... on the big maps.
//
//   gw[k][ci][co] = sum over the pairs (i, o) of kernel offset k of  in[i][ci] * gout[o][co]
//
// Why not k_wgrad_ps.  The position-stationary kernel keeps the accumulators of ALL 27 offsets of a 32 x 96 channel slice in
// the registers of a workgroup, so a 512 x 512 layer is 16 x 6 such slices and every slice re-gathers its rows: at this width
// the launch is made of gather instructions (PMC round 2: 83 M vector-memory instructions for 68 M MFMAs; 15.0 ms = 480
// TFLOP/s).  Here each offset is ONE dense GEMM over its COMPACTED pair list,
//     gw[k] (Cin x Cout) = A_k^T (Cin x M_k) . B_k (M_k x Cout),   A_k = in[pair.in], B_k = gout[pair.out],
// so nothing is padded (the output-stationary forward multiplies 1.3 x the real pairs) and both operands are useful bytes:
//   1. compaction (three small kernels per call, ~0.15 ms at 1.2 M voxels): per (offset, 256-position tile) counts ->
//      exclusive scan per offset -> (in row, out row) lists in position order (deterministic: the fp32 summation order of the
//      result never depends on timing);
//   2. k_wgrad_wide: workgroup = (offset k, range of 16384 POSITIONS, 256 x 256 tile of gw[k]), 8 waves as 2 (ci) x 4 (co), a wave
//      owns 128 x 64 = 4 x 2 MFMA tiles (32x32x16 bf16) in FIXED accumulation registers a[0:127].  A stage = 64 pairs: the two
//      gathered operand tiles ([64 rows][512 B], 32 KB each) come in by LDS-DMA (waves 0..3: `in` rows, waves 4..7: `gout`
//      rows, complementary schedules on each SIMD as in k_conv_wide), double-buffered, one barrier per stage; both MFMA
//      operands are K-major in memory (K = pair index), so the fragments are read with ds_read_b64_tr_b16 (hardware
//      transpose), two per fragment, into fixed VGPRs v[80:127] (the halves of a 4-register fragment cannot be named through
//      inline-asm operands); 64-byte channel segments of a row are XOR-swizzled with (row & 3) on the DMA source so that
//      the four rows a transpose-read touches sit in different bank groups;
//   3. k_wgrad_wide_reduce: the per-range partial tiles (fp32) are added in range order.
// Ranges are ranges of output POSITIONS, the same for all 27 offsets (a range's pairs of offset k are the list entries between
// the scanned tile offsets of its first and last 256-position tile): the 27 workgroups of a range then read the same gout rows
// and neighbouring in rows, so co-resident workgroups share them in L2 / Infinity Cache (pair-index ranges, round 3's first
// version, drift apart across offsets: 8.30 -> 7.51 ms at 1.2 M voxels 512 x 512).  Pair counts stay on the device (no host
// sync): the grid is (ranges x 27 x tiles), a range without pairs writes a zero tile, the reduction adds all of them.
#include "lgs_common.h"

#include <stdlib.h>

namespace lgs {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LGS_AS3(p) ((__attribute__((address_space(3))) void *)(p))
#define LGS_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14)))

constexpr int kWwTile = 256;             // positions per compaction tile
constexpr int kWwStage = 64;             // pairs per stage
constexpr int kWwOp = 32768;             // bytes of one operand tile [64][512 B]
constexpr int kWwLds = 4 * kWwOp;        // [A0][A1][B0][B1]

// ------------------------------------------------------------------------------------------------ pair compaction
__global__ __launch_bounds__(256) void k_ww_count(View v, int ntile, int32_t *__restrict__ cnt) {
  const int tile = blockIdx.x;
  const int64_t p = (int64_t)tile * kWwTile + threadIdx.x;
  for (int k = 0; k < 27; ++k) {
    const bool ok = v.nbr[(int64_t)k * v.n_pad + p] >= 0;
    const int c = __syncthreads_count(ok);
    if (threadIdx.x == 0) cnt[k * ntile + tile] = c;
  }
}

// one block per offset: exclusive scan of the tile counts; total[k] = pairs of offset k
__global__ __launch_bounds__(256) void k_ww_scan(const int32_t *__restrict__ cnt, int ntile, int32_t *__restrict__ off,
                                                 int32_t *__restrict__ total) {
  __shared__ int32_t part[256];
  const int k = blockIdx.x, t = threadIdx.x;
  const int per = (ntile + 255) / 256, b = t * per, e = min(b + per, ntile);
  int32_t s = 0;
  for (int i = b; i < e; ++i) s += cnt[k * ntile + i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int32_t run = 0;
    for (int i = 0; i < 256; ++i) { const int32_t x = part[i]; part[i] = run; run += x; }
    total[k] = run;
  }
  __syncthreads();
  int32_t run = part[t];
  for (int i = b; i < e; ++i) { off[k * ntile + i] = run; run += cnt[k * ntile + i]; }
}

__global__ __launch_bounds__(256) void k_ww_write(View v, int ntile, const int32_t *__restrict__ off, int64_t stride,
                                                  int32_t *__restrict__ pin, int32_t *__restrict__ pout) {
  __shared__ int32_t wbase[4];
  const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t p = (int64_t)tile * kWwTile + threadIdx.x;
  const int32_t orow = v.out_row ? v.out_row[p] : (p < v.n_out ? (int32_t)p : -1);
  for (int k = 0; k < 27; ++k) {
    const int32_t ir = v.nbr[(int64_t)k * v.n_pad + p];
    const bool ok = ir >= 0;
    const uint64_t b = __ballot(ok);
    if (lane == 0) wbase[wave] = __builtin_popcountll(b);
    __syncthreads();
    int32_t base = off[k * ntile + tile];
    for (int w = 0; w < wave; ++w) base += wbase[w];
    if (ok) {
      const int64_t j = (int64_t)k * stride + base + __builtin_popcountll(b & ((1ull << lane) - 1ull));
      pin[j] = ir;
      pout[j] = orow;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ the GEMM
#define LGS_WW_ACC_CLOBBER "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
#define LGS_WW_FRAG_CLOBBER "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

// fragment registers of set S (0 / 1): A blocks a = 0..3 -> v[80 + 24 S + 4 a : +3], B blocks b = 0, 1 -> v[96 + 24 S + 4 b : +3]
// reads of k-step KS (16 pairs): rows 16 KS + ..., first half (+0 rows) into the low register pair, second (+4 rows) into the high
#define LGS_WW_READS(S, KS)                                                                                                   \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 0) ":" LGS_WW_R(S, 1) "], %[pa0] offset:" LGS_WW_OFF(KS, 0) "\n\t"                      \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 2) ":" LGS_WW_R(S, 3) "], %[pa0] offset:" LGS_WW_OFF(KS, 1) "\n\t"                      \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 4) ":" LGS_WW_R(S, 5) "], %[pa1] offset:" LGS_WW_OFF(KS, 0) "\n\t"                      \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 6) ":" LGS_WW_R(S, 7) "], %[pa1] offset:" LGS_WW_OFF(KS, 1) "\n\t"                      \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 8) ":" LGS_WW_R(S, 9) "], %[pa2] offset:" LGS_WW_OFF(KS, 0) "\n\t"                      \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 10) ":" LGS_WW_R(S, 11) "], %[pa2] offset:" LGS_WW_OFF(KS, 1) "\n\t"                    \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 12) ":" LGS_WW_R(S, 13) "], %[pa3] offset:" LGS_WW_OFF(KS, 0) "\n\t"                    \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 14) ":" LGS_WW_R(S, 15) "], %[pa3] offset:" LGS_WW_OFF(KS, 1) "\n\t"                    \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 16) ":" LGS_WW_R(S, 17) "], %[pb0] offset:" LGS_WW_OFF(KS, 0) "\n\t"                    \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 18) ":" LGS_WW_R(S, 19) "], %[pb0] offset:" LGS_WW_OFF(KS, 1) "\n\t"                    \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 20) ":" LGS_WW_R(S, 21) "], %[pb1] offset:" LGS_WW_OFF(KS, 0) "\n\t"                    \
  "ds_read_b64_tr_b16 v[" LGS_WW_R(S, 22) ":" LGS_WW_R(S, 23) "], %[pb1] offset:" LGS_WW_OFF(KS, 1) "\n\t"
// D[ci][co] += A (ci x pairs) . B (pairs x co): accumulator tile (a, b) = a[32 a + 16 b : +15]
#define LGS_WW_MFMA1(S, A, B, ACC)                                                                                             \
  "v_mfma_f32_32x32x16_bf16 a[" ACC "], v[" LGS_WW_R(S, A) ":" LGS_WW_R3(S, A) "], v[" LGS_WW_R(S, B) ":" LGS_WW_R3(S, B) "], a[" ACC "]\n\t"
#define LGS_WW_MFMAS(S)                                                                                                        \
  LGS_WW_MFMA1(S, 0, 16, "0:15") LGS_WW_MFMA1(S, 0, 20, "16:31") LGS_WW_MFMA1(S, 4, 16, "32:47") LGS_WW_MFMA1(S, 4, 20, "48:63")   \
  LGS_WW_MFMA1(S, 8, 16, "64:79") LGS_WW_MFMA1(S, 8, 20, "80:95") LGS_WW_MFMA1(S, 12, 16, "96:111") LGS_WW_MFMA1(S, 12, 20, "112:127")
// register numbers as string literals: set S base 80 + 24 S
#define LGS_WW_STR2(x) #x
#define LGS_WW_STR(x) LGS_WW_STR2(x)
#define LGS_WW_R(S, I) LGS_WW_REGS_##S##_##I
#define LGS_WW_R3(S, I) LGS_WW_REGS3_##S##_##I
#define LGS_WW_OFF(KS, H) LGS_WW_OFF_##KS##_##H
// (generated tables: v-register names of both fragment sets, immediate offsets of the four k-steps)
#define LGS_WW_REGS_0_0 "80"
#define LGS_WW_REGS_0_1 "81"
#define LGS_WW_REGS_0_2 "82"
#define LGS_WW_REGS_0_3 "83"
#define LGS_WW_REGS_0_4 "84"
#define LGS_WW_REGS_0_5 "85"
#define LGS_WW_REGS_0_6 "86"
#define LGS_WW_REGS_0_7 "87"
#define LGS_WW_REGS_0_8 "88"
#define LGS_WW_REGS_0_9 "89"
#define LGS_WW_REGS_0_10 "90"
#define LGS_WW_REGS_0_11 "91"
#define LGS_WW_REGS_0_12 "92"
#define LGS_WW_REGS_0_13 "93"
#define LGS_WW_REGS_0_14 "94"
#define LGS_WW_REGS_0_15 "95"
#define LGS_WW_REGS_0_16 "96"
#define LGS_WW_REGS_0_17 "97"
#define LGS_WW_REGS_0_18 "98"
#define LGS_WW_REGS_0_19 "99"
#define LGS_WW_REGS_0_20 "100"
#define LGS_WW_REGS_0_21 "101"
#define LGS_WW_REGS_0_22 "102"
#define LGS_WW_REGS_0_23 "103"
#define LGS_WW_REGS_1_0 "104"
#define LGS_WW_REGS_1_1 "105"
#define LGS_WW_REGS_1_2 "106"
#define LGS_WW_REGS_1_3 "107"
#define LGS_WW_REGS_1_4 "108"
#define LGS_WW_REGS_1_5 "109"
#define LGS_WW_REGS_1_6 "110"
#define LGS_WW_REGS_1_7 "111"
#define LGS_WW_REGS_1_8 "112"
#define LGS_WW_REGS_1_9 "113"
#define LGS_WW_REGS_1_10 "114"
#define LGS_WW_REGS_1_11 "115"
#define LGS_WW_REGS_1_12 "116"
#define LGS_WW_REGS_1_13 "117"
#define LGS_WW_REGS_1_14 "118"
#define LGS_WW_REGS_1_15 "119"
#define LGS_WW_REGS_1_16 "120"
#define LGS_WW_REGS_1_17 "121"
#define LGS_WW_REGS_1_18 "122"
#define LGS_WW_REGS_1_19 "123"
#define LGS_WW_REGS_1_20 "124"
#define LGS_WW_REGS_1_21 "125"
#define LGS_WW_REGS_1_22 "126"
#define LGS_WW_REGS_1_23 "127"
#define LGS_WW_REGS3_0_0 "83"
#define LGS_WW_REGS3_0_4 "87"
#define LGS_WW_REGS3_0_8 "91"
#define LGS_WW_REGS3_0_12 "95"
#define LGS_WW_REGS3_0_16 "99"
#define LGS_WW_REGS3_0_20 "103"
#define LGS_WW_REGS3_1_0 "107"
#define LGS_WW_REGS3_1_4 "111"
#define LGS_WW_REGS3_1_8 "115"
#define LGS_WW_REGS3_1_12 "119"
#define LGS_WW_REGS3_1_16 "123"
#define LGS_WW_REGS3_1_20 "127"
#define LGS_WW_OFF_0_0 "0"
#define LGS_WW_OFF_0_1 "2048"
#define LGS_WW_OFF_1_0 "8192"
#define LGS_WW_OFF_1_1 "10240"
#define LGS_WW_OFF_2_0 "16384"
#define LGS_WW_OFF_2_1 "18432"
#define LGS_WW_OFF_3_0 "24576"
#define LGS_WW_OFF_3_1 "26624"

struct WwArgs {
  const bf16_t *in, *gout;          // [rows][in_ld], [rows][cout]
  const int32_t *pin, *pout;        // [27][stride] compacted pairs
  const int32_t *total;             // [27]
  const int32_t *off;               // [27][ntile] first pair of every compaction tile
  float *partial;                   // [27 * nchunk][ci_pad][co_pad]
  int64_t stride;
  int cin, cout, in_ld, nchunk, chunk, ntile, ci_pad, co_pad, ti, tj;
  unsigned in_bytes, go_bytes;
};

__global__ __launch_bounds__(512, 2) void k_wgrad_wide(WwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2, wq = wave & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  // workgroup order (speed only): the output tiles of one (range, offset) next to each other, then the 27 offsets of the range:
  // workgroups that run at the same time gather the rows of the same stretch of positions (Infinity-Cache / L2 hits)
  // Workgroup b runs on XCD b % 8 (speed only).  Inside every group of 8 T consecutive workgroups, XCD x takes the T tiles of
  // ONE (range, offset) -- they gather the same rows and now share an L2 -- while the groups themselves stay in dispatch
  // order, so every XCD sees the same mix of long and empty ranges.  (Giving each XCD a contiguous run of the whole order
  // ran 2 x slower: the ranges beyond an offset's pair count exit at once and pile up on the last XCDs.)
  const int T = a.ti * a.tj;
  unsigned lid = blockIdx.x;
  {
    const unsigned G = 8u * (unsigned)T, base = blockIdx.x / G * G;
    if (base + G <= gridDim.x) {
      const unsigned w = blockIdx.x - base, xcd = w & 7u, j = w >> 3;
      lid = base + xcd * (unsigned)T + j;
    }
  }
  const int tile = (int)(lid % (unsigned)T), k = (int)((lid / (unsigned)T) % 27u), chunk = (int)(lid / (unsigned)(27 * T));
  const int ti = tile / a.tj, tj = tile % a.tj;
  const int slot = k * a.nchunk + chunk;
  // the workgroup's pairs: those of offset k whose OUTPUT position lies in position range `chunk` (a.chunk positions, a whole
  // number of compaction tiles) -- the 27 offsets of one range gather the same neighbourhood of rows, so the ranges of the
  // workgroups in flight (a few position ranges x 27 offsets x T tiles) fit the Infinity Cache
  const int tpr = a.chunk / kWwTile, t0 = chunk * tpr, t1 = t0 + tpr;
  const int start = a.off[k * a.ntile + t0];
  const int end = t1 < a.ntile ? a.off[k * a.ntile + t1] : a.total[k];
  const int nstage = (end - start + kWwStage - 1) / kWwStage;

  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" ::: LGS_WW_ACC_CLOBBER);

  constexpr unsigned kOOB = 0xfffff000u;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.in), 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.gout), 0, (int)a.go_bytes, 0x00020000);

  // ---- DMA side: waves 0..3 bring the `in` tile (operand A), waves 4..7 the `gout` tile (operand B); wave wq brings rows
  // [16 wq, 16 wq + 16) = 8 instructions of 2 rows x 512 B
  const bool isA = wm == 0;
  const int32_t *plist = (isA ? a.pin : a.pout) + (int64_t)k * a.stride;
  const unsigned row_bytes = (unsigned)(isA ? a.in_ld : a.cout) * 2u;
  const int C = isA ? a.cin : a.cout;
  const int col0 = (isA ? ti : tj) * 256;
  const int half = lane >> 5, piece = lane & 31;
  // source piece of this lane in tile row r: 64-byte segments XOR (r & 3); r = 16 wq + 2 jj + half -> r & 3 = (2 jj + half) & 3
  unsigned cbyte[2];             // byte offset inside the row for jj even / odd, or kOOB beyond the channel count
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int r3 = (2 * par + half) & 3;
    const int sp = piece ^ (r3 << 2);
    const int ch = col0 + sp * 8;
    cbyte[par] = (ch + 8 <= C) ? (unsigned)ch * 2u : kOOB;
  }
  auto load_idx = [&](int s) __attribute__((always_inline)) -> int32_t {
    const int j = start + s * kWwStage + lane;
    return (s < nstage && j < end) ? plist[j] : -1;
  };
  auto issue = [&](int32_t idxv, int buf) __attribute__((always_inline)) {
    char *dst = smem + (isA ? 0 : 2 * kWwOp) + buf * kWwOp + wq * 8192;
#define LGS_WW_DMA(JJ, IMM)                                                                                         \
    {                                                                                                               \
      const int32_t r0 = __builtin_amdgcn_readlane(idxv, 16 * wq + 2 * (JJ)), r1 = __builtin_amdgcn_readlane(idxv, 16 * wq + 2 * (JJ) + 1); \
      const int32_t r = half ? r1 : r0;                                                                             \
      const unsigned cb = cbyte[(JJ) & 1];                                                                          \
      const unsigned off = (r >= 0 && cb != kOOB) ? (unsigned)r * row_bytes + cb : kOOB;                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rs_in : rs_go, LGS_AS3(dst + (JJ) * 1024), 16, off, 0, 0, 0);  \
    }
    LGS_WW_DMA(0, 0) LGS_WW_DMA(1, 0) LGS_WW_DMA(2, 0) LGS_WW_DMA(3, 0) LGS_WW_DMA(4, 0) LGS_WW_DMA(5, 0) LGS_WW_DMA(6, 0) LGS_WW_DMA(7, 0)
#undef LGS_WW_DMA
  };

  // ---- compute side: transpose-read addresses.  16-lane group g = lane >> 4: cb = g & 1 (16-channel half), hh = g >> 1
  // (rows 8 hh ..); lane i = lane & 15 supplies the 8-byte address (row 8 hh + i / 4 [+ 4], channel 32 blk + 16 cb + 4 (i % 4))
  const int g16 = lane >> 4, i16 = lane & 15;
  const unsigned tr_lane = (unsigned)((8 * (g16 >> 1) + (i16 >> 2)) * 512 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2);
  const unsigned rsw = (unsigned)((i16 >> 2) & 3);
  unsigned pa[4], pb[2];
#pragma unroll
  for (int x = 0; x < 4; ++x) pa[x] = lds0 + tr_lane + ((((unsigned)(wm * 4 + x)) ^ rsw) * 64u);
#pragma unroll
  for (int x = 0; x < 2; ++x) pb[x] = lds0 + 2u * kWwOp + tr_lane + ((((unsigned)(wn * 2 + x)) ^ rsw) * 64u);
  // two k-steps (32 pairs) of a stage: reads of the second step are in flight under the MFMAs of the first
  auto halfstage = [&](int buf, int second) __attribute__((always_inline)) {
    const unsigned o = (unsigned)(buf * kWwOp + second * 16384);
    const unsigned a0 = pa[0] + o, a1 = pa[1] + o, a2 = pa[2] + o, a3 = pa[3] + o, b0 = pb[0] + o, b1 = pb[1] + o;
    asm volatile(LGS_WW_READS(0, 0)
                 LGS_WW_READS(1, 1)
                 "s_waitcnt lgkmcnt(12)\n\t"
                 LGS_WW_MFMAS(0)
                 "s_waitcnt lgkmcnt(0)\n\t"
                 LGS_WW_MFMAS(1)
                 :
                 : [pa0] "v"(a0), [pa1] "v"(a1), [pa2] "v"(a2), [pa3] "v"(a3), [pb0] "v"(b0), [pb1] "v"(b1)
                 : "memory", LGS_WW_ACC_CLOBBER, LGS_WW_FRAG_CLOBBER);
  };

  // ---- main loop (stage s in buffer s & 1): [own DMA landed] [barrier] then the multiply of stage s with this wave's DMA of
  // stage s+1 in it: A-loaders issue first, B-loaders between the two halves (complementary on each SIMD)
  int32_t idx_next = load_idx(0);
  issue(idx_next, 0);
  idx_next = load_idx(1);
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    LGS_VMCNT(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool more = s + 1 < nstage;
    if (isA) {
      if (more) issue(idx_next, buf ^ 1);
      idx_next = load_idx(s + 2);
      halfstage(buf, 0);
      halfstage(buf, 1);
    } else {
      halfstage(buf, 0);
      if (more) issue(idx_next, buf ^ 1);
      idx_next = load_idx(s + 2);
      halfstage(buf, 1);
    }
  }
  LGS_VMCNT(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: partial[(k, chunk)][ci][co]; lane holds column co = .. + (lane & 31), rows ci = .. + 8 q + 4 h + i
  float *dst = a.partial + ((int64_t)slot * a.ci_pad + (ti * 256 + wm * 128)) * a.co_pad + tj * 256 + wn * 64 + (lane & 31);
  const int hrow = 4 * (lane >> 5);
#define LGS_WW_STORE(AB, BB, READS)                                                                                       \
  {                                                                                                                       \
    float t[16];                                                                                                          \
    asm volatile(READS                                                                                                    \
                 : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]), "=v"(t[4]), "=v"(t[5]), "=v"(t[6]), "=v"(t[7]), "=v"(t[8]), \
                   "=v"(t[9]), "=v"(t[10]), "=v"(t[11]), "=v"(t[12]), "=v"(t[13]), "=v"(t[14]), "=v"(t[15])                    \
                 : : "memory");                                                                                           \
    float *d = dst + (int64_t)((AB) * 32 + hrow) * a.co_pad + (BB) * 32;                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) d[(int64_t)(8 * (r >> 2) + (r & 3)) * a.co_pad] = t[r];                \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  LGS_WW_STORE(0, 0, "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\t" "v_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t" "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\t" "v_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\t" "v_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\t" "v_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\t" "v_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\t" "v_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15")
  LGS_WW_STORE(0, 1, "v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\t" "v_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\t" "v_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\t" "v_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\t" "v_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\t" "v_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\t" "v_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\t" "v_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31")
  LGS_WW_STORE(1, 0, "v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\t" "v_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\t" "v_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\t" "v_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\t" "v_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\t" "v_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\t" "v_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\t" "v_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47")
  LGS_WW_STORE(1, 1, "v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\t" "v_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\t" "v_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\t" "v_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\t" "v_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\t" "v_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\t" "v_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\t" "v_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63")
  LGS_WW_STORE(2, 0, "v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\t" "v_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\t" "v_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\t" "v_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\t" "v_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\t" "v_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\t" "v_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\t" "v_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79")
  LGS_WW_STORE(2, 1, "v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\t" "v_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\t" "v_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\t" "v_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\t" "v_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\t" "v_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\t" "v_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\t" "v_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95")
  LGS_WW_STORE(3, 0, "v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\t" "v_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99\n\t" "v_accvgpr_read_b32 %4, a100\n\tv_accvgpr_read_b32 %5, a101\n\t" "v_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a103\n\t" "v_accvgpr_read_b32 %8, a104\n\tv_accvgpr_read_b32 %9, a105\n\t" "v_accvgpr_read_b32 %10, a106\n\tv_accvgpr_read_b32 %11, a107\n\t" "v_accvgpr_read_b32 %12, a108\n\tv_accvgpr_read_b32 %13, a109\n\t" "v_accvgpr_read_b32 %14, a110\n\tv_accvgpr_read_b32 %15, a111")
  LGS_WW_STORE(3, 1, "v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\t" "v_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115\n\t" "v_accvgpr_read_b32 %4, a116\n\tv_accvgpr_read_b32 %5, a117\n\t" "v_accvgpr_read_b32 %6, a118\n\tv_accvgpr_read_b32 %7, a119\n\t" "v_accvgpr_read_b32 %8, a120\n\tv_accvgpr_read_b32 %9, a121\n\t" "v_accvgpr_read_b32 %10, a122\n\tv_accvgpr_read_b32 %11, a123\n\t" "v_accvgpr_read_b32 %12, a124\n\tv_accvgpr_read_b32 %13, a125\n\t" "v_accvgpr_read_b32 %14, a126\n\tv_accvgpr_read_b32 %15, a127")
#undef LGS_WW_STORE
}

// gw[k][ci][co] = sum over the ranges of offset k (in order) of the partial tiles
__global__ __launch_bounds__(256) void k_wgrad_wide_reduce(const float *__restrict__ partial, const int32_t *__restrict__ total,
                                                           int nchunk, int chunk, int ci_pad, int co_pad, int cin, int cout,
                                                           float *__restrict__ gw) {
  const int cq = cout / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)27 * cin * cq;
  if (idx >= tot) return;
  const int q = (int)(idx % cq), ci = (int)((idx / cq) % cin), k = (int)(idx / ((int64_t)cq * cin));
  (void)total; (void)chunk;
  const int nr = nchunk;                       // every position range wrote its tile (zeros when it has no pair of offset k)
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < nr; ++c) {
    const float4 x = *reinterpret_cast<const float4 *>(partial + (((int64_t)(k * nchunk + c) * ci_pad + ci) * co_pad + q * 4));
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  *reinterpret_cast<float4 *>(gw + ((int64_t)k * cin + ci) * cout + q * 4) = s;
}

// ------------------------------------------------------------------------------------------------ host side
struct WwPlan {
  bool ok = false;
  int ntile = 0, nchunk = 0, chunk = 0, ti = 0, tj = 0;
  int64_t cnt_b = 0, pair_b = 0, partial_b = 0, total_b = 0;
};
inline WwPlan ww_plan(const View &v, int cin, int cout) {
  WwPlan p;
  const bool on = tune(T_WGRAD_WIDE) != 0;   // A/B knob
  // maps below ~200 k positions keep the position-stationary kernel (level 2, 81 k rows, 256 -> 256: 0.82 vs 0.31 ms); the
  // parity tests lower WW_MIN_ROWS to send in-network layers of a 70 k-voxel scene through this kernel
  if (!on || v.K != 27 || v.KS != 27 || v.nbr == nullptr || v.n_pad < tune(T_WW_MIN_ROWS) || v.n_pad % kWwTile != 0) return p;
  // positions per workgroup range (a whole number of 256-position tiles): small enough that the rows of the ranges in flight
  // fit the Infinity Cache, large enough that the fp32 partial tiles (256 KB per workgroup) stay a small part of the traffic
  const int range_env = (int)tune(T_WW_RANGE);   // tuning knob
  p.chunk = range_env >= 256 ? range_env / 256 * 256 : 16384;
  if (cin < 256 || cout < 256 || cin % 8 != 0 || cout % 8 != 0) return p;
  while ((int64_t)27 * ((v.n_pad + p.chunk - 1) / p.chunk) * (((cin + 255) / 256) * 256) * (int64_t)(((cout + 255) / 256) * 256) * 4 > (3ll << 30))
    p.chunk *= 2;
  p.ntile = (int)(v.n_pad / kWwTile);
  p.nchunk = (int)((v.n_pad + p.chunk - 1) / p.chunk);
  p.ti = (cin + 255) / 256; p.tj = (cout + 255) / 256;
  p.cnt_b = align256((int64_t)27 * p.ntile * 4);
  p.pair_b = align256((int64_t)27 * v.n_pad * 4);
  p.partial_b = (int64_t)27 * p.nchunk * (p.ti * 256) * (int64_t)(p.tj * 256) * 4;
  p.total_b = 2 * p.cnt_b + 256 + 2 * p.pair_b + align256(p.partial_b);
  p.ok = p.partial_b <= (3ll << 30);
  return p;
}

int64_t wgrad_wide_workspace_bytes(const View &v, int cin, int cout) {
  const WwPlan p = ww_plan(v, cin, cout);
  return p.ok ? p.total_b + 256 : 0;
}

int conv_wgrad_wide(const View &v, const void *in, int cin, int in_ld, const void *gout, int cout, float *gw, void *workspace,
                    hipStream_t s, bool *done) {
  *done = false;
  const WwPlan p = ww_plan(v, cin, cout);
  if (!p.ok) return 0;
  const int ld = in_ld > 0 ? in_ld : cin;
  const uint64_t in_b = (uint64_t)v.n_in * ld * 2, go_b = (uint64_t)v.n_out * cout * 2;
  if (!(in_b < 0xfffff000ull && go_b < 0xfffff000ull) || (ld * 2) % 16 != 0) return 0;
  char *ws = reinterpret_cast<char *>(workspace);
  int32_t *cnt = reinterpret_cast<int32_t *>(ws);
  int32_t *off = reinterpret_cast<int32_t *>(ws + p.cnt_b);
  int32_t *total = reinterpret_cast<int32_t *>(ws + 2 * p.cnt_b);
  int32_t *pin = reinterpret_cast<int32_t *>(ws + 2 * p.cnt_b + 256);
  int32_t *pout = reinterpret_cast<int32_t *>(ws + 2 * p.cnt_b + 256 + p.pair_b);
  float *partial = reinterpret_cast<float *>(ws + 2 * p.cnt_b + 256 + 2 * p.pair_b);
  View vv = v; vv.mirror = 0;
  LGS_KLAUNCH(k_ww_count, dim3(p.ntile), dim3(256), 0, s, vv, p.ntile, cnt);
  LGS_KLAUNCH(k_ww_scan, dim3(27), dim3(256), 0, s, cnt, p.ntile, off, total);
  LGS_KLAUNCH(k_ww_write, dim3(p.ntile), dim3(256), 0, s, vv, p.ntile, off, (int64_t)v.n_pad, pin, pout);
  static bool attr_set = false;
  if (!attr_set) {
    LGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_wide), hipFuncAttributeMaxDynamicSharedMemorySize, kWwLds));
    attr_set = true;
  }
  WwArgs a;
  a.in = reinterpret_cast<const bf16_t *>(in); a.gout = reinterpret_cast<const bf16_t *>(gout);
  a.pin = pin; a.pout = pout; a.total = total; a.off = off; a.ntile = p.ntile; a.partial = partial; a.stride = v.n_pad;
  a.cin = cin; a.cout = cout; a.in_ld = ld; a.nchunk = p.nchunk; a.chunk = p.chunk; a.ci_pad = p.ti * 256; a.co_pad = p.tj * 256; a.ti = p.ti; a.tj = p.tj;
  a.in_bytes = (unsigned)in_b; a.go_bytes = (unsigned)go_b;
  LGS_KLAUNCH(k_wgrad_wide, dim3((unsigned)(27 * p.nchunk * p.ti * p.tj)), dim3(512), kWwLds, s, a);
  const int64_t tot = (int64_t)27 * cin * (cout / 4);
  LGS_KLAUNCH(k_wgrad_wide_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, partial, total, p.nchunk, p.chunk, a.ci_pad, a.co_pad,
                     cin, cout, gw);
  LGS_HIP(hipGetLastError());
  *done = true;
  return 0;
}

}  // namespace lgs
