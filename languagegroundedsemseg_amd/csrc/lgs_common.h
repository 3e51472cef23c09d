// lgs_common.h -- internal helpers shared by the engine's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/lgs_engine.h"

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

}  // namespace lgs

struct lgs_kmap {
  lgs_manager *mgr = nullptr;
  int in_key = -1, out_key = -1, ks = 0, K = 1;
  lgs::View fwd;  // gathers from the in map, writes the out map
  lgs::View bwd;  // gathers from the out map, writes the in map (dgrad / transposed conv)
};
