// lgs_tuning.hip -- ONE table of the engine's tuning / debugging knobs, and the per-launch-site dispatch counters.
//
// Every knob used to be a `static const ... getenv(...)` next to its launch site (20 of them by round 3).  They now live in
// the table below: name, default, what it does.  A knob's initial value comes from the environment variable LGS_<NAME> (read
// once, at the first query), lgs_tuning_set() changes it at run time (the parity tests use that, e.g. to send a 70 k-row
// layer through the kernel that production only picks above 200 k rows), lgs_tuning_describe() prints the table.
//
// Dispatch counters: every kernel launch of the engine goes through LGS_KLAUNCH (lgs_common.h), which bumps a counter keyed
// by the launch site's kernel expression + the template bindings of the enclosing function.  tests/ read them to assert that
// every kernel instantiation the benchmarked steps dispatch is also dispatched by a parity test (round 3 shipped a hot
// kernel that no test reached because of a size gate).
#include <atomic>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lgs_common.h"

namespace lgs {

namespace {
struct Knob { Tune id; const char *name; int64_t def; const char *doc; };
// in the order of enum Tune (lgs_common.h): every row names its enumerator, init_knobs() refuses a table whose row i is not knob i
const Knob kKnobs[T_COUNT] = {
    {T_WW_MIN_ROWS, "WW_MIN_ROWS", 200000, "k_wgrad_wide (per-offset GEMM over compacted pair lists) is used for 3^3 weight gradients with >= 256 x 256 "
                            "channels on maps of at least this many positions; below, k_wgrad_ps wins (81 k rows, 256->256: 0.82 vs 0.31 ms)"},
    {T_WW_RANGE, "WW_RANGE", 16384, "positions per k_wgrad_wide workgroup range (multiple of 256; 8192 overflows the partial-tile cap, 65536: 8.6 vs 7.5 ms)"},
    {T_WGRAD_WIDE, "WGRAD_WIDE", 1, "0 = never use k_wgrad_wide (A/B)"},
    {T_BN_FUSED, "BN_FUSED", 1, "0 = BatchNorm always as three launches (column sums, fold, apply) instead of the persistent grid-barrier kernels; "
                    "required when several processes time-share one GPU"},
    {T_BN_FUSED_MAX_MB, "BN_FUSED_MAX_MB", 24, "layers above this size use the three-launch BatchNorm (they are bandwidth-bound; the 4096-workgroup apply streams faster)"},
    {T_BN_FUSED_FWD_MAX_MB, "BN_FUSED_FWD_MAX_MB", 0, "the same bound for the FORWARD direction alone; 0 = the forward is always three launches.  Round 4: with the host "
                               "no longer the bound the one-launch forward loses at every size (8-scene step 27.93 vs 28.15 ms, one scene 10.17 vs "
                               "10.30; stand-alone 20 MB: 22.8 vs 33.0 us), the one-launch BACKWARD still wins (16.4 vs 16.8 ms of backward)"},
    {T_BN_FUSED_BLOCKS, "BN_FUSED_BLOCKS", 0, "workgroups of the fused BatchNorm kernels; 0 = min(256, co-resident workgroups of the device)"},
    {T_PS_CUS, "PS_CUS", 20, "CUs per XCD that k_wgrad_ps fills (of 32): the compute stream keeps CUs of its own next to the CU-owning weight gradient"},
    {T_PS_WIDE3, "PS_WIDE3", 1, "k_wgrad_ps: three-block stationary slices for >= 256-channel layers (0 = two-block)"},
    {T_WGRAD_PS, "WGRAD_PS", 1, "0 = force the round-1 pair-list weight gradient (k_wgrad_bf16) everywhere (debugging)"},
    {T_MASK_WINDOW, "MASK_WINDOW", 16384, "3^3 maps: positions per window inside which rows are re-sorted by their 27-bit neighbour mask (1024 .. 65536 swept)"},
    {T_CONV_SPLIT, "CONV_SPLIT", 1, "0 = no slot split (3 x 9 offsets into fp32 partial images) for under-filled coarse-level 3^3 launches"},
    {T_SMALL_CFG, "SMALL_CFG", 0, "k_conv_gather tile for maps < 65536 positions: 0 = automatic (id 8), 5 / 9 / 10 / 11 = the other measured shapes"},
    {T_WIDE_GC64, "WIDE_GC64", 1, "k_conv_wide: 64-channel stages per reduction group"},
    {T_ARENA_DBG, "ARENA_DBG", 0, "print coordinate-manager arena allocations to stderr"},
    {T_CONV_WIDE, "CONV_WIDE", 1, "0 = >= 256-output-channel layers stay on k_conv_gather's 8-wave tile (id 16) instead of k_conv_wide (A/B)"},
    {T_MASK_ORDER, "MASK_ORDER", 0, "3^3 maps: sort code of the neighbourhood mask inside a window: 0 the mask, 1 corners > edges > faces, 2 faces > edges > corners, 3 popcount-major"},
    {T_BN_FOLD, "BN_FOLD", 1, "1 = BatchNorm layers up to BN_FOLD_MAX_MB run as TWO launches per direction: column sums into <= BN_FOLD_PARTS partial rows, "
                   "then an apply kernel whose workgroups fold the partial rows themselves (no fold launch, no grid barrier); 0 = the round-4 "
                   "policies (grid-barrier kernel backward, three launches forward)"},
    {T_BN_FOLD_MAX_MB, "BN_FOLD_MAX_MB", 6, "layers above this size keep the round-4 policies.  Stand-alone (tools/dbg/bn_small.py, bf16): 2.6 MB 15.2 / 21.1 us forward / backward against 16.1 / 29.0 (three launches / grid barriers), 5 MB 17.8 / 26.0 against 16.0 / 27.6; at 10 - 20 MB the wider 512-row reduction wins (22 / 37 - 59 against 16 - 23 / 32 - 46)"},
    {T_BN_FOLD_PARTS, "BN_FOLD_PARTS", 64, "partial rows of the two-launch BatchNorm (<= 64): fewer = a shorter fold in every apply workgroup, more = a wider reduction"},
    {T_BN_FOLD_GRID, "BN_FOLD_GRID", 256, "apply workgroups of the two-launch BatchNorm (each folds the partial rows redundantly)"},
    {T_FP32_SPLIT, "FP32_SPLIT", 1, "fp32 tensors: 1 = convolution forward / dgrad multiply on the bf16 matrix pipe with exactly split operands (x = hi + mid + lo, "
                      "six bf16 products accumulated in fp32: error below 2^-22 |x||w|, 2.7 x the MFMA rate of v_mfma_f32_32x32x2_f32); "
                      "0 = the exact-fp32 MFMA instance"},
    {T_BLOCK_WGRAD_LATE, "BLOCK_WGRAD_LATE", 0, "A/B: 1 = lgs_block_backward forks the FIRST convolution's weight gradient after the block's last dgrad instead of "
                            "beside it (the 96->128 dgrad of block8.0 runs 2.1 x its stand-alone time next to k_wgrad_ps 128->96)"},
    {T_WGRAD_F32_LDS, "WGRAD_F32_LDS", 2, "fp32 weight gradients: 2 = rows staged through LDS as three exactly-split bf16 planes, six bf16 MFMAs per 16 pairs "
                                          "(k_wgrad_f32s_lds; dropped terms < 2^-24 |x g|, fp32 accumulate).  Stand-alone it wins at >= 96 input channels only "
                                          "(96 -> 96: 3.06 vs 3.38 ms; 32 -> 32: 1.06 vs 0.84), in the STEP it wins everywhere: 82.4 - 82.9 vs 86.1 - 86.7 ms -- it "
                                          "holds the matrix pipe 2.7 x shorter than the exact-fp32 MFMA and the convolutions on the other stream get it "
                                          "(round 6); 3 = split only where it also wins stand-alone (83.5 - 83.7 ms); 1 = exact-fp32 MFMA on staged rows "
                                          "(k_wgrad_f32_lds; bit-identical to 0); 0 = k_wgrad_f32 (one 4-byte load per lane and MFMA operand)"},
    {T_WIDE_SCHED, "WIDE_SCHED", 0, "k_conv_wide: where the two waves of a SIMD issue the next stage's LDS-DMA between their four row blocks.  0 = "
                                    "weights first / gathers after block 1 (round 3); 1 = gathers first, weights after block 1; 2 = gathers first, "
                                    "weights after block 0; 3 = gathers first, weights after block 2; 4 = both first (round 2's schedule)"},
    {T_HEAD_TILE, "HEAD_TILE", 0, "1x1 layers with 5 - 7 output blocks on the big maps (the 200-class head): 0 = one 128-position tile spans all output "
                                  "channels (the feature matrix is streamed once; 7 x 16 accumulator registers per wave), 1 = the generic 4-block tile "
                                  "(two column tiles, features read twice, 3 waves per SIMD)"},
    {T_POINTWISE, "POINTWISE", 1, "bf16 1x1 stride-1 convolutions (and their dgrads) on maps of >= 65536 positions with <= 224 output channels run on "
                                  "k_pointwise (persistent streaming GEMM, lgs_pointwise.hip) instead of k_conv_gather's identity-map tiles where that wins (5 - 7 output "
                                  "blocks, or > 128 reduction channels into 3 blocks: the 200-class head and its dgrad); 2 = every shape it serves; 0 = off (A/B)"},
};
std::atomic<int64_t> g_val[T_COUNT];
std::once_flag g_once;

void init_knobs() {
  for (int i = 0; i < T_COUNT; ++i) {
    if ((int)kKnobs[i].id != i) {     // a knob added to the enum and the table in different places (round 5 shipped one for an hour)
      fprintf(stderr, "lgs_engine: tuning table row %d is %s but enum Tune says otherwise -- fix csrc/lgs_tuning.hip\n", i, kKnobs[i].name);
      abort();
    }
    const std::string env = std::string("LGS_") + kKnobs[i].name;
    const char *e = getenv(env.c_str());
    g_val[i].store(e ? atoll(e) : kKnobs[i].def, std::memory_order_relaxed);
  }
}
int find_knob(const char *name) {
  if (!name) return -1;
  if (!strncmp(name, "LGS_", 4)) name += 4;
  for (int i = 0; i < T_COUNT; ++i)
    if (!strcmp(name, kKnobs[i].name)) return i;
  return -1;
}

// ---- dispatch counters
constexpr int kMaxSites = 1024;
struct Site { std::string name; std::atomic<int64_t> hits{0}; };
Site g_sites[kMaxSites];
int g_nsites = 0;
std::mutex g_site_mu;

std::string squeeze(const char *s) {   // drop blanks and the outer parentheses hipLaunchKernelGGL needs around template commas
  std::string o;
  for (; *s; ++s)
    if (*s != ' ' && *s != '\t' && *s != '\n') o.push_back(*s);
  while (o.size() >= 2 && o.front() == '(' && o.back() == ')') o = o.substr(1, o.size() - 2);
  return o;
}
}  // namespace

int64_t tune(Tune t) {
  std::call_once(g_once, init_knobs);
  return g_val[t].load(std::memory_order_relaxed);
}

int dispatch_site(const char *kernel_text, const char *pretty_function) {
  std::string name = squeeze(kernel_text);
  // template bindings of the enclosing function, e.g. "[T = unsigned short, KIND = 0]" (clang's __PRETTY_FUNCTION__)
  if (pretty_function) {
    const char *b = strrchr(pretty_function, '[');
    if (b && strchr(b, '=')) name += " " + squeeze(b);
  }
  std::lock_guard<std::mutex> lk(g_site_mu);
  for (int i = 0; i < g_nsites; ++i)
    if (g_sites[i].name == name) return i;
  if (g_nsites >= kMaxSites) return kMaxSites - 1;
  g_sites[g_nsites].name = name;
  return g_nsites++;
}

void dispatch_hit(int site) { g_sites[site].hits.fetch_add(1, std::memory_order_relaxed); }

}  // namespace lgs

extern "C" {

int lgs_tuning_set(const char *name, int64_t value) {
  std::call_once(lgs::g_once, lgs::init_knobs);
  const int i = lgs::find_knob(name);
  LGS_REQUIRE(i >= 0, std::string("lgs_tuning_set: unknown knob '") + (name ? name : "(null)") + "'");
  lgs::g_val[i].store(value, std::memory_order_relaxed);
  return 0;
}

int lgs_tuning_get(const char *name, int64_t *value) {
  std::call_once(lgs::g_once, lgs::init_knobs);
  const int i = lgs::find_knob(name);
  LGS_REQUIRE(i >= 0 && value, std::string("lgs_tuning_get: unknown knob '") + (name ? name : "(null)") + "'");
  *value = lgs::g_val[i].load(std::memory_order_relaxed);
  return 0;
}

// "NAME\tdefault\tvalue\tdoc\n" per knob; returns the bytes needed (incl. the terminating NUL); writes at most cap bytes
int64_t lgs_tuning_describe(char *buf, int64_t cap) {
  std::call_once(lgs::g_once, lgs::init_knobs);
  std::string s;
  for (int i = 0; i < lgs::T_COUNT; ++i)
    s += std::string(lgs::kKnobs[i].name) + "\t" + std::to_string(lgs::kKnobs[i].def) + "\t" +
         std::to_string(lgs::g_val[i].load()) + "\t" + lgs::kKnobs[i].doc + "\n";
  if (buf && cap > 0) {
    const int64_t n = std::min<int64_t>(cap - 1, (int64_t)s.size());
    memcpy(buf, s.data(), (size_t)n);
    buf[n] = 0;
  }
  return (int64_t)s.size() + 1;
}

// "count\tsite\n" per launch site that was hit since the last reset; same size protocol as lgs_tuning_describe
int64_t lgs_debug_dispatch_counts(char *buf, int64_t cap, int reset) {
  std::string s;
  {
    std::lock_guard<std::mutex> lk(lgs::g_site_mu);
    for (int i = 0; i < lgs::g_nsites; ++i) {
      const int64_t h = reset ? lgs::g_sites[i].hits.exchange(0) : lgs::g_sites[i].hits.load();
      if (h > 0) s += std::to_string(h) + "\t" + lgs::g_sites[i].name + "\n";
    }
  }
  if (buf && cap > 0) {
    const int64_t n = std::min<int64_t>(cap - 1, (int64_t)s.size());
    memcpy(buf, s.data(), (size_t)n);
    buf[n] = 0;
  }
  return (int64_t)s.size() + 1;
}

}  // extern "C"
