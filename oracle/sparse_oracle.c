/*
 * sparse_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A slow, obviously-correct restatement of generalized sparse convolution as the
 * reference's callers use it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (languagegroundedsemseg_amd/)
 * never does.
 *
 * PARITY STATUS: the arithmetic of this path lives in MinkowskiEngine==0.5.4
 * (pinned at /root/reference/config/lg_semseg.yml:204), which is NOT vendored under
 * /root/reference and is not installable here.  This oracle is therefore
 * "parity unpinned" against MinkowskiEngine itself.  It is pinned instead against an
 * independent known-answer source: dense torch conv3d / conv_transpose3d on densified
 * grids (tests/test_oracle_dense.py), as SURVEY.md section 8c prescribes.
 *
 * What it restates (reference call sites):
 *   - coordinate dedup, first occurrence wins            SparseTensor(feats, coords)
 *         /root/reference/lib/train_test/pl_BaselineTrainer.py:300
 *   - stride-2 coarsening floor(c / ts) * ts             conv(kernel_size=2, stride=2)
 *         /root/reference/models/res16unet.py:49-56,66-73,83-90,100-107
 *   - kernel maps for HYPER_CUBE regions, ks in {1,2,3}   models/modules/common.py:179-236
 *   - conv / transposed conv forward, dgrad, wgrad        MinkowskiConvolution[Transpose]
 *         /root/reference/models/modules/common.py:195-203,228-236
 *
 * Conventions (shared with the HIP engine, see include/lgs_engine.h):
 *   coords  : int32 [N,4] = (batch, x, y, z)            lib/transforms.py:421
 *   weights : float [K, Cin, Cout]; offset index k enumerates the hypercube with the
 *             first spatial axis fastest.  Odd sizes are centred
 *             (k = (dx+1) + 3(dy+1) + 9(dz+1)), even sizes one-sided (k = dx+2dy+4dz).
 *   kernel map = triples (k, in_row, out_row) with  c_in = c_out + off_k * ts_in
 *             (ts_in = tensor stride of the input map).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ hash set of 4-int coords */
typedef struct {
  int64_t cap;          /* power of two */
  int64_t *slot;        /* row index or -1 */
  const int32_t *coords;/* borrowed [N,4] */
} cmap_t;

static uint64_t mix4(const int32_t *c) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (int d = 0; d < 4; ++d) {
    h ^= (uint64_t)(uint32_t)c[d] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
  }
  return h;
}

static int cmap_init(cmap_t *m, const int32_t *coords, int64_t n) {
  int64_t cap = 16;
  while (cap < 2 * n + 2) cap <<= 1;
  m->cap = cap;
  m->coords = coords;
  m->slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
  if (!m->slot) return -1;
  for (int64_t i = 0; i < cap; ++i) m->slot[i] = -1;
  return 0;
}
static void cmap_free(cmap_t *m) { free(m->slot); m->slot = NULL; }

/* returns existing row for key c, or inserts `row` and returns it */
static int64_t cmap_find_or_insert(cmap_t *m, const int32_t *c, int64_t row) {
  uint64_t i = mix4(c) & (uint64_t)(m->cap - 1);
  for (;;) {
    int64_t r = m->slot[i];
    if (r < 0) { m->slot[i] = row; return row; }
    if (memcmp(m->coords + 4 * r, c, 16) == 0) return r;
    i = (i + 1) & (uint64_t)(m->cap - 1);
  }
}
static int64_t cmap_find(const cmap_t *m, const int32_t *c) {
  uint64_t i = mix4(c) & (uint64_t)(m->cap - 1);
  for (;;) {
    int64_t r = m->slot[i];
    if (r < 0) return -1;
    if (memcmp(m->coords + 4 * r, c, 16) == 0) return r;
    i = (i + 1) & (uint64_t)(m->cap - 1);
  }
}

static int32_t floor_div(int32_t a, int32_t b) { /* b > 0, floor toward -inf */
  int32_t q = a / b, r = a % b;
  return (r != 0 && r < 0) ? q - 1 : q;
}

/* ------------------------------------------------------------------ coordinate maps */

/* Dedup: first occurrence wins, unique_index ascending.
 * unique_index[n_unique] (caller allocs N), inverse[N]. Returns n_unique. */
int64_t orc_unique_coords(const int32_t *coords, int64_t n, int64_t *unique_index, int64_t *inverse) {
  cmap_t m;
  if (cmap_init(&m, coords, n)) return -1;
  int64_t *row2u = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  int64_t nu = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t r = cmap_find_or_insert(&m, coords + 4 * i, i);
    if (r == i) { row2u[i] = nu; unique_index[nu++] = i; }
    inverse[i] = row2u[r];
  }
  free(row2u);
  cmap_free(&m);
  return nu;
}

/* Coarsen: out = unique( floor(c / ts_out) * ts_out ) per batch, first-occurrence order.
 * out_coords caller-allocated [N,4]; parent[N] = out row of each input row. Returns n_out. */
int64_t orc_stride_coords(const int32_t *coords, int64_t n, int32_t ts_out, int32_t *out_coords, int64_t *parent) {
  cmap_t m;
  if (cmap_init(&m, out_coords, n)) return -1;
  int64_t no = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t c[4];
    c[0] = coords[4 * i];
    for (int d = 1; d < 4; ++d) c[d] = floor_div(coords[4 * i + d], ts_out) * ts_out;
    memcpy(out_coords + 4 * no, c, 16); /* tentative append so the map can compare against it */
    int64_t r = cmap_find_or_insert(&m, c, no);
    if (r == no) ++no;
    parent[i] = r;
  }
  cmap_free(&m);
  return no;
}

static void offset_of(int k, int ks, int32_t scale, int32_t off[3]) {
  /* first spatial axis fastest; odd centred, even one-sided */
  int base = (ks % 2 == 1) ? -(ks / 2) : 0;
  for (int d = 0; d < 3; ++d) { off[d] = (base + k % ks) * scale; k /= ks; }
}

/* Kernel map: all (k, in_row, out_row) with c_in = c_out + off_k * ts_in.
 * Triples are emitted k-major (all pairs of k=0, then k=1, ...), out_row ascending within k.
 * Arrays caller-allocated with capacity n_out * ks^3. Returns M (number of pairs). */
int64_t orc_kernel_map(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords, int64_t n_out,
                       int ks, int32_t ts_in, int32_t *km_k, int64_t *km_in, int64_t *km_out) {
  cmap_t m;
  if (cmap_init(&m, in_coords, n_in)) return -1;
  for (int64_t i = 0; i < n_in; ++i) cmap_find_or_insert(&m, in_coords + 4 * i, i);
  int K = ks * ks * ks;
  int64_t M = 0;
  for (int k = 0; k < K; ++k) {
    int32_t off[3];
    offset_of(k, ks, ts_in, off);
    for (int64_t o = 0; o < n_out; ++o) {
      int32_t c[4] = {out_coords[4 * o], out_coords[4 * o + 1] + off[0], out_coords[4 * o + 2] + off[1],
                      out_coords[4 * o + 3] + off[2]};
      int64_t i = cmap_find(&m, c);
      if (i >= 0) { km_k[M] = k; km_in[M] = i; km_out[M] = o; ++M; }
    }
  }
  cmap_free(&m);
  return M;
}

/* ------------------------------------------------------------------ convolution arithmetic
 * All three are pair-list driven:  out[o] += in[i] . W[k]   for every triple (k,i,o).
 * A transposed convolution uses the same routines with the map's in/out columns swapped
 * by the caller (ME reuses the forward map of the matching strided conv, Appendix A).
 * Accumulation is in double so the oracle is the "truth" side of a 1e-3 fp32 comparison. */

static void csr_by(const int64_t *key, int64_t M, int64_t n, int64_t **ptr_out, int64_t **perm_out) {
  int64_t *ptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
  int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
  for (int64_t p = 0; p < M; ++p) ptr[key[p] + 1]++;
  for (int64_t r = 0; r < n; ++r) ptr[r + 1] += ptr[r];
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  memcpy(cur, ptr, sizeof(int64_t) * (size_t)n);
  for (int64_t p = 0; p < M; ++p) perm[cur[key[p]]++] = p;
  free(cur);
  *ptr_out = ptr; *perm_out = perm;
}

/* out[n_out,Cout] = sum over pairs in[i,:] @ W[k]  (+ bias[Cout] if non-NULL) */
void orc_conv_forward(const float *in, int64_t n_in, int cin, const float *w, int cout, const float *bias,
                      const int32_t *km_k, const int64_t *km_in, const int64_t *km_out, int64_t M,
                      float *out, int64_t n_out) {
  (void)n_in;
  int64_t *ptr, *perm;
  csr_by(km_out, M, n_out, &ptr, &perm);
#pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * (size_t)cout);
#pragma omp for schedule(dynamic, 64)
    for (int64_t o = 0; o < n_out; ++o) {
      for (int c = 0; c < cout; ++c) acc[c] = bias ? (double)bias[c] : 0.0;
      for (int64_t q = ptr[o]; q < ptr[o + 1]; ++q) {
        int64_t p = perm[q];
        const float *x = in + km_in[p] * cin;
        const float *wk = w + (int64_t)km_k[p] * cin * cout;
        for (int ci = 0; ci < cin; ++ci) {
          double xv = x[ci];
          const float *wr = wk + (int64_t)ci * cout;
          for (int c = 0; c < cout; ++c) acc[c] += xv * (double)wr[c];
        }
      }
      for (int c = 0; c < cout; ++c) out[o * cout + c] = (float)acc[c];
    }
    free(acc);
  }
  free(ptr); free(perm);
}

/* gin[n_in,Cin] = sum over pairs gout[o,:] @ W[k]^T */
void orc_conv_dgrad(const float *gout, int64_t n_out, int cout, const float *w, int cin,
                    const int32_t *km_k, const int64_t *km_in, const int64_t *km_out, int64_t M,
                    float *gin, int64_t n_in) {
  (void)n_out;
  int64_t *ptr, *perm;
  csr_by(km_in, M, n_in, &ptr, &perm);
#pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * (size_t)cin);
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_in; ++i) {
      for (int c = 0; c < cin; ++c) acc[c] = 0.0;
      for (int64_t q = ptr[i]; q < ptr[i + 1]; ++q) {
        int64_t p = perm[q];
        const float *g = gout + km_out[p] * cout;
        const float *wk = w + (int64_t)km_k[p] * cin * cout;
        for (int ci = 0; ci < cin; ++ci) {
          const float *wr = wk + (int64_t)ci * cout;
          double s = 0.0;
          for (int c = 0; c < cout; ++c) s += (double)g[c] * (double)wr[c];
          acc[ci] += s;
        }
      }
      for (int c = 0; c < cin; ++c) gin[i * cin + c] = (float)acc[c];
    }
    free(acc);
  }
  free(ptr); free(perm);
}

/* gw[K,Cin,Cout] = sum over pairs of k:  in[i,:]^T (outer) gout[o,:] */
void orc_conv_wgrad(const float *in, int cin, const float *gout, int cout, int K,
                    const int32_t *km_k, const int64_t *km_in, const int64_t *km_out, int64_t M, float *gw) {
  int64_t per = (int64_t)cin * cout;
#pragma omp parallel for schedule(dynamic, 1)
  for (int k = 0; k < K; ++k) {
    double *acc = (double *)calloc((size_t)per, sizeof(double));
    for (int64_t p = 0; p < M; ++p) {
      if (km_k[p] != k) continue;
      const float *x = in + km_in[p] * cin;
      const float *g = gout + km_out[p] * cout;
      for (int ci = 0; ci < cin; ++ci) {
        double xv = x[ci];
        double *ar = acc + (int64_t)ci * cout;
        for (int c = 0; c < cout; ++c) ar[c] += xv * (double)g[c];
      }
    }
    for (int64_t e = 0; e < per; ++e) gw[k * per + e] = (float)acc[e];
    free(acc);
  }
}

int orc_version(void) { return 1; }
