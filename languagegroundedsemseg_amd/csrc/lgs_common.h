// lgs_common.h -- internal helpers shared by the engine's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/lgs_engine.h"

struct lgs_kmap;

namespace lgs {

void set_error(const std::string &msg);

#define LGS_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      lgs::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                     std::to_string(__LINE__) + ")");                                              \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

#define LGS_REQUIRE(cond, msg)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      lgs::set_error(std::string(msg) + " [" #cond "] (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

// ---- tuning table + dispatch counters (lgs_tuning.hip)
enum Tune {
  T_WW_MIN_ROWS, T_WW_RANGE, T_WGRAD_WIDE, T_BN_FUSED, T_BN_FUSED_MAX_MB, T_BN_FUSED_FWD_MAX_MB, T_BN_FUSED_BLOCKS, T_PS_CUS, T_PS_WIDE3, T_WGRAD_PS,
  T_MASK_WINDOW, T_CONV_SPLIT, T_SMALL_CFG, T_WIDE_GC64, T_ARENA_DBG, T_CONV_WIDE, T_MASK_ORDER,
  T_BN_FOLD, T_BN_FOLD_MAX_MB, T_BN_FOLD_PARTS, T_BN_FOLD_GRID, T_FP32_SPLIT, T_BLOCK_WGRAD_LATE, T_WGRAD_F32_LDS, T_WIDE_SCHED, T_HEAD_TILE, T_POINTWISE,
  T_COUNT
};
int64_t tune(Tune t);                                                   // current value (environment LGS_<NAME> at start, lgs_tuning_set later)
int dispatch_site(const char *kernel_text, const char *pretty_function);   // registers a launch site once -> its index
void dispatch_hit(int site);
// every kernel launch of the engine: counted per launch site (kernel expression + template bindings of the enclosing function)
#define LGS_KLAUNCH(kernel, ...)                                                                 \
  do {                                                                                           \
    static const int _lgs_site = lgs::dispatch_site(#kernel, __PRETTY_FUNCTION__);               \
    lgs::dispatch_hit(_lgs_site);                                                                \
    hipLaunchKernelGGL(kernel, __VA_ARGS__);                                                     \
  } while (0)

constexpr int kPadRows = 256;  // every position array is padded to a multiple of this
constexpr int kGroup = 64;     // rows per mask / tile_k entry (one wavefront of positions)

inline int64_t pad_rows(int64_t n) { return (n + kPadRows - 1) / kPadRows * kPadRows; }

// A view of a kernel map as an OUTPUT-STATIONARY gather table:
// position p in [0,n_pad) produces output row out_row[p] (or p itself) as
//     out[row(p)] = sum over slots s with nbr[s][p] >= 0 of  in[nbr[s][p]] . W[weight_index(s, p)]
struct View {
  const int32_t *nbr = nullptr;      // [KS][n_pad] input row per (slot, position), -1 = none; NULL = identity (1x1)
  const uint32_t *mask64 = nullptr;  // [n_pad/64] bit s set if any of the 64 positions has slot s   (KS > 1)
  const int32_t *tile_k = nullptr;   // [n_pad/64] weight index of the single slot, -1 = empty group (KS == 1, grouped)
  const int32_t *out_row = nullptr;  // [n_pad] output row per position, -1 = padding; NULL = identity
  int64_t n_pad = 0;
  int64_t n_out = 0;  // rows of the tensor this view writes
  int64_t n_in = 0;   // rows of the tensor it gathers from
  int KS = 1;         // slots per position (27 / 8 / 1)
  int K = 1;          // weight matrices of the op (27 / 8 / 1)
  int mirror = 0;     // weight index = K-1-s (the 3^3 map read in the dgrad direction)
};

inline int pad32(int c) { return (c + 31) / 32 * 32; }
inline int esize(int dtype) { return dtype == LGS_BF16 ? 2 : 4; }
inline int epl(int dtype) { return dtype == LGS_BF16 ? 8 : 4; }
inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

#if defined(__HIPCC__)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef uint16_t bf16_t;  // storage type tag for bf16 tensors
__device__ inline float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ inline uint16_t f32_to_bf16(float f) {  // round to nearest even
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ inline float ld_elem(const float *p) { return *p; }
__device__ inline float ld_elem(const bf16_t *p) { return bf16_to_f32(*p); }
#endif

int64_t wgrad_workspace_bytes(const lgs_kmap *km, int cin, int cout, int dtype);
// lgs_conv_wide.hip: 2-D blocked forward / dgrad for >= 256 output channels (bf16); same packed weight image as k_conv_gather's wide tile
// 1x1 stride-1 convolutions of the big maps as a streaming GEMM (lgs_pointwise.hip)
bool pointwise_supported(const View &v, int K, int g_real, int o_real, int64_t in_ld);
int launch_pointwise(const View &v, const void *in, int64_t in_ld, int g_real, const float *w, int cin_w, int cout_w, int transposed,
                     int o_real, const float *bias, void *out, hipStream_t s);
bool pointwise_f32_supported(const View &v, int K, int g_real, int o_real, int64_t in_ld);
int launch_pointwise_f32(const View &v, const void *in, int64_t in_ld, int g_real, const float *w, int cin_w, int cout_w, int transposed,
                         int o_real, const float *bias, void *out, hipStream_t s);
int launch_conv_wide(const View &v, const void *in, int cin_real, int in_ld, const void *wp, int nb_total, int ncp, int nbp, int K,
                     void *out, int cout_real, const float *bias, int gc64, hipStream_t s);
// lgs_wgrad_wide.hip: per-offset dense GEMM over compacted pair lists for >= 256 x 256 channel 3^3 weight gradients (bf16)
int64_t wgrad_wide_workspace_bytes(const View &v, int cin, int cout);
int conv_wgrad_wide(const View &v, const void *in, int cin, int in_ld, const void *gout, int cout, float *gw, void *workspace,
                    hipStream_t s, bool *done);
// order `stream` after the construction of km's manager's maps (they are built on the manager's own stream)
int kmap_wait(lgs_kmap *km, hipStream_t stream);

}  // namespace lgs

struct lgs_kmap {
  lgs_manager *mgr = nullptr;
  int in_key = -1, out_key = -1, ks = 0, K = 1;
  lgs::View fwd;  // gathers from the in map, writes the out map
  lgs::View bwd;  // gathers from the out map, writes the in map (dgrad / transposed conv)
};
