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
#include <string.h>

#include "lgs_common.h"

namespace lgs {

namespace {
struct Knob { const char *name; int64_t def; const char *doc; };
// keep in the order of enum Tune (lgs_common.h)
const Knob kKnobs[T_COUNT] = {
    {"WW_MIN_ROWS", 200000, "k_wgrad_wide (per-offset GEMM over compacted pair lists) is used for 3^3 weight gradients with >= 256 x 256 "
                            "channels on maps of at least this many positions; below, k_wgrad_ps wins (81 k rows, 256->256: 0.82 vs 0.31 ms)"},
    {"WW_RANGE", 16384, "positions per k_wgrad_wide workgroup range (multiple of 256; 8192 overflows the partial-tile cap, 65536: 8.6 vs 7.5 ms)"},
    {"WGRAD_WIDE", 1, "0 = never use k_wgrad_wide (A/B)"},
    {"BN_FUSED", 1, "0 = BatchNorm always as three launches (column sums, fold, apply) instead of the persistent grid-barrier kernels; "
                    "required when several processes time-share one GPU"},
    {"BN_FUSED_MAX_MB", 24, "layers above this size use the three-launch BatchNorm (they are bandwidth-bound; the 4096-workgroup apply streams faster)"},
    {"BN_FUSED_FWD_MAX_MB", 0, "the same bound for the FORWARD direction alone; 0 = the forward is always three launches.  Round 4: with the host "
                               "no longer the bound the one-launch forward loses at every size (8-scene step 27.93 vs 28.15 ms, one scene 10.17 vs "
                               "10.30; stand-alone 20 MB: 22.8 vs 33.0 us), the one-launch BACKWARD still wins (16.4 vs 16.8 ms of backward)"},
    {"BN_FUSED_BLOCKS", 0, "workgroups of the fused BatchNorm kernels; 0 = min(256, co-resident workgroups of the device)"},
    {"PS_CUS", 20, "CUs per XCD that k_wgrad_ps fills (of 32): the compute stream keeps CUs of its own next to the CU-owning weight gradient"},
    {"PS_WIDE3", 1, "k_wgrad_ps: three-block stationary slices for >= 256-channel layers (0 = two-block)"},
    {"WGRAD_PS", 1, "0 = force the round-1 pair-list weight gradient (k_wgrad_bf16) everywhere (debugging)"},
    {"MASK_WINDOW", 16384, "3^3 maps: positions per window inside which rows are re-sorted by their 27-bit neighbour mask (1024 .. 65536 swept)"},
    {"CONV_SPLIT", 1, "0 = no slot split (3 x 9 offsets into fp32 partial images) for under-filled coarse-level 3^3 launches"},
    {"SMALL_CFG", 0, "k_conv_gather tile for maps < 65536 positions: 0 = automatic (id 8), 5 / 9 / 10 / 11 = the other measured shapes"},
    {"WIDE_GC64", 1, "k_conv_wide: 64-channel stages per reduction group"},
    {"WIDE_DBG", 0, "k_conv_wide knock-out bits for time attribution (RESULTS ARE WRONG): 1 no LDS reads / MFMA, 4 no gathers, 8 no weight DMA"},
    {"WIDE_TRACE", 0, "k_conv_wide: print per-phase shader-clock sums of one workgroup (debug instance of the kernel)"},
    {"ARENA_DBG", 0, "print coordinate-manager arena allocations to stderr"},
    {"CONV_WIDE", 1, "0 = >= 256-output-channel layers stay on k_conv_gather's 8-wave tile (id 16) instead of k_conv_wide (A/B)"},
    {"MASK_ORDER", 0, "3^3 maps: sort code of the neighbourhood mask inside a window: 0 the mask, 1 corners > edges > faces, 2 faces > edges > corners, 3 popcount-major"},
};
std::atomic<int64_t> g_val[T_COUNT];
std::once_flag g_once;

void init_knobs() {
  for (int i = 0; i < T_COUNT; ++i) {
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
