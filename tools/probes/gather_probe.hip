// probe: cost of a row gather on gfx950 as a function of the LANE -> (row, 16-byte segment) mapping.
//   mode 0  the MFMA operand layout: lane (vx = l % 32, h = l / 32) reads 16 B of row vx -> the 4 lanes of a quad touch
//           4 different rows (what k_conv_gather did up to round 2)
//   mode 1  row-contiguous: lane l reads segment l % 4 of row l / 4 -> a quad covers 64 contiguous bytes of ONE row
//   mode 2/3  the same rows through global_load (missing neighbours -> row 0 / predicated off)
//   mode 4  reference point: coalesced 1 KB loads from a 32 KB (L1-resident) region
//   mode 5/6  8 / 4 bytes per lane instead of 16
// usage: gather_probe.bin <channels> <1 = rows in spatial order, 0 = random> <KB of LDS per workgroup: 70 -> 2 workgroups
//        per CU, 1 -> registers decide> <0 = ~half of the 27 neighbours present, 1 = all, 2 = a quarter in whole 64-row groups>
// Same bytes per wave instruction (1 KB), same rows, same number of instructions; 8 waves / CU like the conv kernel
// (dynamic LDS limits the occupancy).  Prints ms per pass and GB/s of gathered (non-missing) bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned kOOB = 0xfffff000u;

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint16_t *in, int64_t in_bytes, const int32_t *nbr, int64_t n_pad, int row_bytes,
                                                 int nchunk, uint32_t *out) {
  extern __shared__ char lds_dummy[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t pos = (int64_t)blockIdx.x * 256 + wave * 64;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(in), 0, (int)in_bytes, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  for (int k = 0; k < 27; ++k) {
    const int32_t *ix = nbr + (int64_t)k * n_pad + pos;
    // per-lane row indices for the 4 loads of a chunk
    int32_t r[4];
    if (MODE != 1) {
      const int vx = lane & 31;
      r[0] = r[1] = ix[vx];
      r[2] = r[3] = ix[32 + vx];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ix[j * 16 + (lane >> 2)];
    }
    unsigned base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned seg;
      if (MODE != 1) seg = (lane >> 5) * 32 + (j & 1) * 16;
      else seg = (lane & 3) * 16;
      base[j] = r[j] >= 0 ? (unsigned)r[j] * (unsigned)row_bytes + seg : kOOB;
    }
    {
      for (int c = 0; c < nchunk; ++c) {
        u32x4 f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE <= 1) f[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base[j] == kOOB ? kOOB : base[j] + c * 64, 0, 0);
          else if (MODE == 2) f[j] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(in) + (base[j] == kOOB ? 0u : base[j] + c * 64));   // global_load, missing -> row 0
          else if (MODE == 3) { f[j] = u32x4{0, 0, 0, 0}; if (base[j] != kOOB) f[j] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(in) + base[j] + c * 64); }   // predicated
          else if (MODE == 4) f[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((k * 3 + c) * 4 + j) & 31) * 1024u + lane * 16u, 0, 0);   // coalesced, 32 KB region
          else if (MODE == 5) { auto t2 = __builtin_amdgcn_raw_buffer_load_b64(rs, base[j] == kOOB ? kOOB : base[j] + c * 64, 0, 0); f[j] = u32x4{t2[0], t2[1], 0, 0}; }
          else if (MODE == 6) { f[j] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rs, base[j] == kOOB ? kOOB : base[j] + c * 64, 0, 0), 0, 0, 0}; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= f[j];
      }
    }
  }
  out[(int64_t)blockIdx.x * 256 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main(int argc, char **argv) {
  const int64_t n = 1200000, n_pad = (n + 255) / 256 * 256;
  const int C = argc > 1 ? atoi(argv[1]) : 96;
  const int row_bytes = C * 2, nchunk = C / 32;
  const int sorted = argc > 2 ? atoi(argv[2]) : 1;
  const int pmode = argc > 4 ? atoi(argv[4]) : 0;   // 0: ~half of the neighbours, scattered; 1: all present; 2: a quarter, whole 64-row groups present or absent
  // a synthetic "surface": 27 offsets, each present with probability ~0.5, neighbours at small / medium / large row distances
  std::vector<int32_t> nbr(27 * n_pad, -1);
  std::vector<int32_t> perm(n);
  for (int64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
  srand(1);
  if (!sorted) for (int64_t i = n - 1; i > 0; --i) { int64_t j = ((int64_t)rand() * 32768 + rand()) % (i + 1); std::swap(perm[i], perm[j]); }
  const int d1[3] = {-1, 0, 1}, d2[3] = {-350, 0, 350}, d3[3] = {-90000, 0, 90000};
  int64_t live = 0;
  for (int k = 0; k < 27; ++k) {
    const int dd = d1[k % 3] + d2[(k / 3) % 3] + d3[k / 9];
    for (int64_t p = 0; p < n; ++p) {
      // presence pattern constant over runs of 64 positions for ~half of the offsets, random for the others
      const bool present = pmode == 1 ? true : pmode == 2 ? (k % 2 == 0 && (p / 64) % 2 == 0) || k == 13 : k == 13 || ((k % 3 != 1 || (k / 9) == 1) ? ((p / 64 * 2654435761u + k * 40503u) >> 7 & 3) != 0 && (rand() & 3) != 0 : (rand() & 7) < 2);
      const int64_t q = p + dd;
      if (present && q >= 0 && q < n) { nbr[(int64_t)k * n_pad + p] = perm[q]; ++live; }
    }
  }
  printf("rows %ld, %d B/row, %.2f neighbours per row, input order %s\n", (long)n, row_bytes, (double)live / n, sorted ? "spatial" : "random");
  uint16_t *in; int32_t *dn; uint32_t *out;
  CK(hipMalloc(&in, n * row_bytes)); CK(hipMemset(in, 1, n * row_bytes));
  CK(hipMalloc(&dn, nbr.size() * 4)); CK(hipMemcpy(dn, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, n_pad * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(n_pad / 256);
  const size_t lds = (argc > 3 ? atoi(argv[3]) : 70) * 1024;   // 70 KB: 2 workgroups per CU
  typedef void (*kern_t)(const uint16_t *, int64_t, const int32_t *, int64_t, int, int, uint32_t *);
  kern_t kerns[7] = {k_gather<0>, k_gather<1>, k_gather<2>, k_gather<3>, k_gather<4>, k_gather<5>, k_gather<6>};
  const char *names[7] = {"buffer b128, MFMA lane layout", "buffer b128, row-contiguous quads", "global b128, missing -> row 0", "global b128, predicated",
                          "buffer b128 coalesced 32 KB region", "buffer b64 gather", "buffer b32 gather"};
  for (int mode = 0; mode < 7; ++mode) {
    CK(hipFuncSetAttribute((const void *)kerns[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kerns[mode], dim3(grid), dim3(256), lds, 0, in, n * row_bytes, dn, n_pad, row_bytes, nchunk, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      if (rep) printf("mode %d (%s): %.3f ms per pass, %.1f ns per wave load instruction per CU\n", mode, names[mode], ms,
                      ms * 1e6 / ((double)(n_pad / 64) * 27 * nchunk * 4 / 256));
    }
  }
  return 0;
}
