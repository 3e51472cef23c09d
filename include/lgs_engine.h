/*
 * lgs_engine.h -- C-ABI of the MI355X-native sparse-voxel engine (liblgs_engine.so).
 *
 * This is the drop-in boundary UNDER the MinkowskiEngine Python operator surface that
 * RozDavid/LanguageGroundedSemseg is written against.  The reference never binds native code
 * for this path itself -- it calls MinkowskiEngine==0.5.4 (config/lg_semseg.yml:204) through
 * Python -- so every entry point below cites the Python call site(s) of the reference whose
 * work it performs.  The Python host code in languagegroundedsemseg_amd/me/ keeps the ME names
 * (SparseTensor, MinkowskiConvolution, ...) and calls these functions through ctypes.
 *
 * Conventions
 *   - plain C types only; no torch types.  All `const void*` / `void*` buffers are DEVICE
 *     pointers owned by the caller (torch) and only borrowed for the duration of the call.
 *   - compute entry points (conv / bn / ce / clip) enqueue on the HIP stream passed in `stream`
 *     (a hipStream_t cast to void*; NULL = default stream) and never synchronise.
 *   - coordinate-manager entry points build their maps on the manager's OWN stream; `stream` is the
 *     caller's stream, used only for ordering (inputs produced on it are waited for, outputs written
 *     to caller memory are published to it).  lgs_manager_insert and lgs_kmap_export synchronise the
 *     manager's stream once to return a row count to the host (never the caller's compute backlog);
 *     the insert also counts the rows of the eight coarser levels, so lgs_manager_stride2 returns its
 *     count WITHOUT a synchronisation (one host sync per input batch instead of five).  Every compute call that takes an lgs_kmap
 *     orders itself after the manager's map work with a stream-side event wait (no host sync).
 *   - return value: 0 = OK, non-zero = error; lgs_last_error() returns the message of the
 *     last failing call on this thread.  The Python side raises RuntimeError with it.
 *   - a manager and everything it owns is not thread-safe; one manager per input batch
 *     (ME semantics: maps/kernel maps are cached for exactly one forward+backward).
 *   - coords are int32 [N,4] = (batch, x, y, z), batch in column 0
 *     (/root/reference/lib/transforms.py:421, lib/train_test/pl_BaselineTrainer.py:294).
 *     Supported range: batch in [0,1024), |x|,|y|,|z| < 131072 (checked; error otherwise).
 *   - weights are float32 [K, Cin, Cout] (ME parameter layout), kernel-offset index
 *     k enumerates the hypercube with the first spatial axis fastest; odd sizes centred, even
 *     sizes one-sided.  1x1 convs use K = 1.
 *   - features are row-major [N, C] in `dtype` (LGS_F32 or LGS_BF16); accumulation is fp32.
 */
#ifndef LGS_ENGINE_H
#define LGS_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGS_ABI_VERSION 13

enum lgs_dtype { LGS_F32 = 0, LGS_BF16 = 1 };

typedef struct lgs_manager lgs_manager; /* coordinate manager: owns coordinate maps + kernel maps */
typedef struct lgs_kmap lgs_kmap;       /* one cached kernel map (owned by its manager) */

/* ---- library ---------------------------------------------------------------------------- */
int lgs_abi_version(void);
const char *lgs_last_error(void);

/* ---- tuning table and dispatch counters (csrc/lgs_tuning.hip) -----------------------------
 * No counterpart in the reference (MinkowskiEngine exposes no such hooks): this is the engine's ONE table of tuning and
 * debugging knobs (which used to be scattered getenv calls) and the test hook the parity suite needs to prove coverage.
 *   lgs_tuning_set / get : knob by name ("WW_MIN_ROWS" or "LGS_WW_MIN_ROWS"); initial value = environment LGS_<NAME>, else the
 *                          default.  Unknown name -> error.  Knobs take effect at the next launch.
 *   lgs_tuning_describe  : "NAME\tdefault\tvalue\tdoc\n" per knob into buf (at most cap bytes incl. NUL); returns bytes needed.
 *   lgs_debug_dispatch_counts : "count\tlaunch site\n" for every kernel launch site hit since the last reset (site = kernel
 *                          expression + template bindings, e.g. "k_wgrad_ps<KIND,NCS> [KIND=0,NCS=3]"); reset != 0 zeroes them.
 *                          tests/test_gpu_dispatch_coverage.py: every site the benchmarked steps of bench.py dispatch
 *                          (/root/reference/scripts/train_models.sh, text_representation_train.sh) must also be dispatched
 *                          by a parity test.                                                  */
int lgs_tuning_set(const char *name, int64_t value);
int lgs_tuning_get(const char *name, int64_t *value);
int64_t lgs_tuning_describe(char *buf, int64_t cap);
int64_t lgs_debug_dispatch_counts(char *buf, int64_t cap, int reset);

/* ---- coordinate manager ------------------------------------------------------------------
 * replaces: ME.SparseTensor(features, coordinates) -> CoordinateManager.insert_and_map
 *   /root/reference/lib/train_test/pl_BaselineTrainer.py:300
 *   /root/reference/lib/train_test/pl_RepresentationTrainer.py:183
 *   /root/reference/downstream/insseg/lib/pl_Trainer.py:263                                  */
int lgs_manager_create(int device, lgs_manager **out);
int lgs_manager_destroy(lgs_manager *mgr);
/* Insert coords[N,4] as the tensor-stride-1 map.  Dedups (first occurrence wins, surviving
 * rows keep input order).  Writes the map key to *key and the unique-row count to *n_unique
 * (host; this call synchronises `stream` once).
 * unique_index (device int64[N], first n_unique valid) and inverse (device int64[N]) may be NULL. */
int lgs_manager_insert(lgs_manager *mgr, const int32_t *coords, int64_t n, int64_t *unique_index,
                       int64_t *inverse, void *stream, int *key, int64_t *n_unique);

/* Coarsen map `in_key` by 2 per axis: unique floor(c / 2ts) * 2ts per batch.  Reuses the map if it
 * already exists (ME: one map per tensor stride).  Synchronises `stream` once when it creates one.
 * replaces: the output-coordinate generation inside conv(kernel_size=2, stride=2)
 *   /root/reference/models/res16unet.py:49-56,66-73,83-90,100-107                              */
int lgs_manager_stride2(lgs_manager *mgr, int in_key, void *stream, int *out_key, int64_t *n_out);
/* Device-side consistency flags of this manager's maps, read back with ONE synchronisation of the manager's map stream
 * (test / debug infrastructure: the map builders size coarse maps from counts taken at insert time and never synchronise;
 * a builder that finds its own count disagreeing raises the flag instead of writing past its arrays).  *flags: 0 = consistent,
 * bit 0 = coordinate out of the key range at insert, bit 1 = a coarse map's row count differs from the insert-time count.
 * Same call sites as lgs_manager_stride2.                                                                             */
int lgs_manager_check(lgs_manager *mgr, int *flags);

/* Finer map that `key` was coarsened from (-1 if none); used by transposed convs to land on the
 * cached map (/root/reference/models/res16unet.py:116-124 + me.cat at :237). */
int lgs_manager_parent_of(lgs_manager *mgr, int key, int *fine_key);

int lgs_manager_map_size(lgs_manager *mgr, int key, int64_t *n, int *tensor_stride);
/* coords of map `key` into dst (device int32 [n,4]); replaces SparseTensor.C (pl_BaselineTrainer.py:384) */
int lgs_manager_get_coords(lgs_manager *mgr, int key, int32_t *dst, void *stream);

/* Kernel map between two maps of this manager (cached per (in_key,out_key,kernel_size)).
 *   kernel_size 3, in_key == out_key            : 3x3x3 stride-1 (a4)
 *   kernel_size 2, out_key == stride2(in_key)   : 2x2x2 stride-2 (a5); the transposed conv (a6)
 *                                                 uses the same object with `transposed` views
 *   kernel_size 1, in_key == out_key            : identity (a7)
 * replaces: the implicit kernel-map construction in MinkowskiConvolution[Transpose].forward
 *   /root/reference/models/modules/common.py:179-236                                           */
int lgs_manager_kernel_map(lgs_manager *mgr, int in_key, int out_key, int kernel_size, void *stream,
                           lgs_kmap **out);

/* Export the map as (k, in_row, out_row) triples for set-equality parity tests.
 * Pass NULL buffers to query *m only (synchronises). Buffers are device int32[*m]. */
int lgs_kmap_export(lgs_kmap *km, int32_t *k, int32_t *in_row, int32_t *out_row, void *stream, int64_t *m);

/* ---- sparse convolution --------------------------------------------------------------------
 * replaces MinkowskiConvolution / MinkowskiConvolutionTranspose forward + autograd backward
 *   /root/reference/models/modules/common.py:195-203 (conv), :228-236 (conv_tr)
 *   call sites /root/reference/models/res16unet.py:196-270, models/modules/resnet_block.py:41-57
 *
 * `transposed` = 0: forward direction of the map (in rows -> out rows);
 *                1: the transposed convolution (map's out rows -> map's in rows).
 * `weight` is always the module's own parameter [K,Cin,Cout] (float32), Cin/Cout being THIS
 * op's input/output channels.  `workspace` must hold lgs_conv_workspace_bytes(...) bytes.
 * Zero-copy ME.cat (/root/reference/models/res16unet.py:237,247,257,267): both inputs of a concat are written straight
 * into the concat buffer (lgs_bn_forward's y_row_stride), so the SKIP half is afterwards a column slice of a wider
 * row-major tensor; its readers take a row stride: lgs_conv_forward / lgs_conv_wgrad `in_row_stride` (the strided conv that
 * consumes the skip tensor; bf16 only for wgrad) and lgs_bn_backward `y_row_stride` (the ReLU mask of the norm that
 * produced it).  Rows must stay 16-byte aligned. */
int64_t lgs_conv_workspace_bytes(const lgs_kmap *km, int cin, int cout, int dtype, int op /*0 fwd,1 dgrad,2 wgrad*/);

/* Packed weight images.  The conv kernels read the weights in MFMA fragment order (bf16 / fp32, padded to the tile
 * configuration of the launch shape); by default every call re-packs the fp32 [K,Cin,Cout] parameter into its workspace
 * (~6 us, 125 launches per Res16UNet34C step).  A caller that keeps the image across calls asks lgs_conv_pack_desc for
 * its layout (bytes == 0: this shape packs internally only), owns a device buffer of `bytes`, and passes it as `packed`:
 *   pack_mode 1 = pack into it now, then run;  2 = it is up to date, run straight away;  (packed == NULL: mode 0, internal).
 * lgs_pack_weights_batch re-packs ANY number of images in one launch (descs_device = device copy of the descriptors
 * with `weight` / `packed` filled in, max_total = largest `total`): the host wrapper runs it once after the optimiser
 * step, so the convolutions of the next step find their images ready. */
typedef struct lgs_pack_desc {
  const float *weight; /* device float32 [K, cin_w, cout_w] */
  void *packed;        /* device buffer of `bytes` */
  int64_t bytes, total;
  int K, cin_w, cout_w, transposed, mirror, g_real, o_real, ncp, nbp, dtype;
} lgs_pack_desc;
int lgs_conv_pack_desc(const lgs_kmap *km, int op /* 0 forward, 1 dgrad */, int transposed, int cin, int cout, int dtype,
                       lgs_pack_desc *out);
int lgs_pack_weights_batch(const lgs_pack_desc *descs_device, int n, int64_t max_total, void *stream);

/* out[n_out,cout] = conv(in[n_in,cin]) (+ bias[cout] if non-NULL)
 * bn_partial (may be NULL): the BatchNorm that follows the conv in every block of the model family
 *   (/root/reference/models/modules/resnet_block.py:41-57: conv -> norm) needs sum / sum of squares of this output per
 *   channel; the conv epilogue can emit them per position tile -- float32 [rows][2][cout], rows =
 *   lgs_conv_bn_partial_rows(...) (0 = this launch shape cannot, pass NULL) -- as sum(y - pivot), sum((y - pivot)^2) of the
 *   STORED values, so lgs_bn_forward / lgs_bn_stats need not read the output again for their statistics pass.
 *   bn_pivot: float32 [cout] per-channel shift (BatchNorm's running mean), NULL = 0. */
int lgs_conv_bn_partial_rows(const lgs_kmap *km, int transposed, int cout, int dtype);
int lgs_conv_forward(lgs_kmap *km, int transposed, const void *in, int cin, const float *weight, int cout,
                     const float *bias, void *out, int dtype, void *workspace, float *bn_partial, const float *bn_pivot,
                     void *packed, int pack_mode, int in_row_stride /* elements; 0 = cin */, void *stream);
/* grad_in[n_in,cin] from grad_out[n_out,cout] */
int lgs_conv_dgrad(lgs_kmap *km, int transposed, const void *grad_out, int cout, const float *weight, int cin,
                   void *grad_in, int dtype, void *workspace, void *packed, int pack_mode, void *stream);

/* grad_in += dgrad(grad_out): the sum autograd forms when the convolution's input also feeds a residual branch
 * (models/modules/resnet_block.py:41-57: `out += residual`), taken in the kernel epilogue and rounded exactly like "store the
 * dgrad, then add the two tensors".  lgs_conv_dgrad_can_accumulate() tells whether the launch shape of (km, transposed, cin,
 * cout, dtype) has that epilogue (1) or the caller has to add the tensors itself (0); lgs_conv_dgrad_accumulate() fails
 * for a shape that has not. */
int lgs_conv_dgrad_can_accumulate(const lgs_kmap *km, int transposed, int cin, int cout, int dtype);
int lgs_conv_dgrad_accumulate(lgs_kmap *km, int transposed, const void *grad_out, int cout, const float *weight, int cin,
                              void *grad_in, int dtype, void *workspace, void *packed, int pack_mode, void *stream);
/* grad_weight[K,cin,cout] (float32, overwritten) */
int lgs_conv_wgrad(lgs_kmap *km, int transposed, const void *in, int cin, const void *grad_out, int cout,
                   float *grad_weight, int dtype, void *workspace, int in_row_stride /* elements; 0 = cin */, void *stream);
/* 1 if lgs_conv_wgrad can read `in` through `in_row_stride` for this shape (only the position-stationary bf16 kernel does;
 * it declines e.g. for >= 4 GiB at the wider stride, odd channel counts, or when forced off), else 0: the caller then
 * passes a contiguous copy.  The Python host asks before every strided weight gradient instead of letting the call fail. */
int lgs_conv_wgrad_supports_stride(const lgs_kmap *km, int transposed, int cin, int cout, int dtype, int in_row_stride);

/* ---- fused batch-norm / ReLU / residual ------------------------------------------------------
 * replaces ME.MinkowskiBatchNorm (.bn = nn.BatchNorm1d over all rows) + MinkowskiReLU + `out += residual`
 *   /root/reference/models/modules/common.py:17-19, models/modules/resnet_block.py:41-57
 * Training-mode batch statistics over all n rows.  stats = float32 [2*C] workspace:
 * on return mean[C], invstd[C].  running_mean/var (float32 [C]) updated with `momentum`
 * (unbiased variance), may be NULL; num_batches_tracked (device int64 scalar, nn.BatchNorm1d's buffer) is
 * incremented by the same kernel, may be NULL.  residual may be NULL.  y may alias x.
 * workspace: lgs_bn_workspace_bytes(n, c) bytes of caller-owned device scratch (no allocation inside).
 * conv_partials / conv_partial_rows / pivot: statistics already produced by the preceding lgs_conv_forward (see there);
 * NULL / 0 = compute them from x. */
int64_t lgs_bn_workspace_bytes(int64_t n, int c);
int lgs_bn_forward(const void *x, int64_t n, int c, const float *gamma, const float *beta, float eps,
                   float momentum, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                   const void *residual, int relu, void *y, float *stats, int dtype, void *workspace,
                   const float *conv_partials /* lgs_conv_forward's bn_partial or NULL */, int conv_partial_rows,
                   const float *pivot /* the bn_pivot that conv call was given */, int64_t y_row_stride /* elements; 0 = c */,
                   void *stream);
/* Backward of the fused op.  x = forward input, stats = the forward's mean/invstd.
 * relu: 0 = none; 1 = ReLU mask taken from the forward OUTPUT y (required when a residual was added);
 *       2 = mask recomputed from x as (xhat*gamma + beta > 0), y may be NULL (one tensor read fewer).
 * Writes dx, dgamma[C], dbeta[C]; if dresidual != NULL also the gradient flowing to the residual (= masked dy).
 * dy_row_stride (elements, 0 = C): dy may be a column slice of a wider row-major tensor -- the gradient of one input
 * of ME.cat (res16unet.py:233-262) is exactly that, and reading it in place saves a copy of the whole slice. */
int lgs_bn_backward(const void *x, const void *y, const void *dy, int64_t dy_row_stride, int64_t n, int c,
                    const float *gamma, const float *beta, const float *stats, int relu, void *dx, void *dresidual,
                    float *dgamma, float *dbeta, int dtype, void *workspace, int64_t y_row_stride /* of y; 0 = c */,
                    void *stream);

/* The same op in halves, so that data-parallel training can exchange the statistics between ranks in the middle
 * (ME.MinkowskiSyncBatchNorm, /root/reference/main.py:122-123) with one small collective per direction and NO host-side
 * tensor arithmetic in between:
 *   lgs_bn_stats           -> rec[2C+1] = local mean[C], local M2[C] (sum of squared deviations from it), row count
 *   (all-gather of the records of all ranks: all_stats[world][2C+1])
 *   lgs_bn_sync_combine    -> stats[2C] = global mean, invstd (Chan's parallel formula, double); updates running_mean /
 *                             running_var / num_batches_tracked (each may be NULL); *inv_n_total = 1 / global rows
 *   lgs_bn_apply           <- stats[2C]
 *   lgs_bn_backward_reduce -> sums[2C] = local sum dy', local sum dy' * xhat (dy' = dy masked by ReLU); the same two
 *                             vectors also go to dgamma / dbeta (may be NULL): parameter gradients stay local
 *   (all-reduce of sums)
 *   lgs_bn_backward_apply  <- sums[2C] (all-reduced), 1/N either by value (inv_n_total) or, if inv_n_device != NULL,
 *                             read from the device scalar lgs_bn_sync_combine wrote (no host sync)
 * y_row_stride / dy_row_stride (elements, 0 = c): y / dy may be column slices of wider row-major buffers (zero-copy ME.cat: the
 * norm writes into, and its backward reads from, the concat buffer), rows 16-byte aligned. */
int lgs_bn_stats(const void *x, int64_t n, int c, float *rec /* [2C+1] */, int dtype, void *workspace,
                 const float *conv_partials, int conv_partial_rows, const float *pivot, void *stream);
int lgs_bn_sync_combine(const float *all_stats, int world, int c, float eps, float momentum, float *running_mean,
                        float *running_var, int64_t *num_batches_tracked, float *stats, float *inv_n_total, void *stream);
int lgs_bn_apply(const void *x, int64_t n, int c, const float *gamma, const float *beta, const float *stats,
                 const void *residual, int relu, void *y, int dtype, int64_t y_row_stride, void *stream);
int lgs_bn_backward_reduce(const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                           const float *beta, const float *stats, int relu, float *sums, float *dgamma, float *dbeta,
                           int dtype, void *workspace, int64_t dy_row_stride, int64_t y_row_stride, void *stream);
int lgs_bn_backward_apply(const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                          const float *beta, const float *stats, const float *sums, float inv_n_total,
                          const float *inv_n_device, int relu, void *dx, void *dresidual, int dtype, int64_t dy_row_stride,
                          int64_t y_row_stride, void *stream);

/* ---- SyncBatchNorm as one call per direction, on the engine's own RCCL communicator (csrc/lgs_comm.hip) --------
 * replaces the per-layer statistics exchange of ME.MinkowskiSyncBatchNorm (convert_sync_batchnorm, /root/reference/main.py:121-123;
 * downstream/insseg/lib/ddp_trainer.py:191-194) when it runs rank per GPU: the split kernels above with ncclAllGather /
 * ncclAllReduce issued ON THE CALLER'S STREAM between them (no process-group stream hand-over, no host code in between).
 *   lgs_comm_unique_id   rank 0: 128 opaque bytes (ncclGetUniqueId) for the ranks to share (e.g. one torch.distributed broadcast)
 *   lgs_comm_create      every rank, collectively (ncclCommInitRank on `device`); lgs_comm_destroy releases it
 *   lgs_bn_forward_sync  = lgs_bn_stats -> all-gather [world][2C+1] -> lgs_bn_sync_combine -> lgs_bn_apply; stats [2C] and
 *                          inv_n [1] (device scalar 1 / global rows) are outputs the backward takes back
 *   lgs_bn_backward_sync = lgs_bn_backward_reduce -> all-reduce [2C] -> lgs_bn_backward_apply (dgamma / dbeta stay local)
 * RCCL is resolved at run time from the librccl the process has loaded; without one lgs_comm_* fail with a message and the
 * caller keeps its own collectives.  workspace: lgs_bn_sync_workspace_bytes(n, c, world). */
typedef struct lgs_comm lgs_comm;
int lgs_comm_unique_id(void *id128);
int lgs_comm_create(const void *id128, int world, int rank, int device, lgs_comm **out);
/* Mailbox mode (no RCCL): lgs_comm_create_ipc allocates this rank's mailbox in device memory and returns its 64-byte
 * hipIpcMemHandle; the ranks exchange the handles out of band (one torch.distributed all-gather) and every rank passes ALL of
 * them ([world][64] bytes, its own entry ignored) to lgs_comm_ipc_open.  lgs_bn_forward_sync / lgs_bn_backward_sync then exchange
 * through ONE kernel each (peer-to-peer stores into every rank's mailbox, arrival flags spun on in the kernel) instead of an RCCL
 * collective.  All ranks of one communicator must issue the same sequence of calls.  Same call sites as above. */
int lgs_comm_create_ipc(int world, int rank, int device, lgs_comm **out, void *handle64);
int lgs_comm_ipc_open(lgs_comm *comm, const void *handles64);
int lgs_comm_destroy(lgs_comm *comm);
int lgs_comm_world(const lgs_comm *comm);
int64_t lgs_bn_sync_workspace_bytes(int64_t n, int c, int world);
int lgs_bn_forward_sync(lgs_comm *comm, const void *x, int64_t n, int c, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                        const void *residual, int relu, void *y, float *stats, float *inv_n, int dtype, void *workspace,
                        int64_t y_row_stride, void *stream);
int lgs_bn_backward_sync(lgs_comm *comm, const void *x, const void *y, const void *dy, int64_t n, int c, const float *gamma,
                         const float *beta, const float *stats, const float *inv_n, int relu, void *dx, void *dresidual,
                         float *dgamma, float *dbeta, int dtype, void *workspace, int64_t dy_row_stride, int64_t y_row_stride,
                         void *stream);

/* ---- one call per residual block and direction (csrc/lgs_block.hip) -----------------------------
 * replaces the call sequence of BasicBlock.forward and of its autograd backward
 *   /root/reference/models/modules/resnet_block.py:41-57   (conv1 - norm1 - relu - conv2 - norm2 - (+ residual) - relu)
 *   /root/reference/models/resnet.py:93-103                 (downsample = 1x1 conv + norm on the residual branch)
 * for batches small enough that the training step is bound by the HOST enqueueing ~250 engine calls (one ~150 k-voxel
 * scene per step).  Exactly the launches of the call-by-call path, in the same order, on the caller's stream: results are
 * bit-identical.  Every pointer is a device pointer owned by the caller; packed weight images / modes as in lgs_conv_forward
 * (forward images for lgs_block_forward, dgrad images for lgs_block_backward).  Training mode only (batch statistics).     */
typedef struct lgs_bn_params {
  const float *gamma, *beta;
  float *running_mean, *running_var;     /* may be NULL together */
  int64_t *num_batches_tracked;          /* may be NULL */
  float eps, momentum;
} lgs_bn_params;
typedef struct lgs_block_fwd {
  lgs_kmap *km3, *km1;                   /* 3^3 map of the level; 1x1 map (NULL = no downsample branch) */
  int dtype, relu_final, cin, planes;
  int64_t n;                             /* rows of the level */
  const void *x;                         /* [n, cin] */
  const float *w1, *w2, *wd;             /* fp32 [27, cin, planes], [27, planes, planes], [cin, planes] or NULL */
  void *pk1, *pk2, *pkd;                 /* packed images (lgs_conv_pack_desc, op 0) or NULL */
  int pm1, pm2, pmd;                     /* pack modes as for lgs_conv_forward */
  lgs_bn_params n1, n2, nd;
  void *o1, *y1, *o2, *od, *res, *y2;    /* [n, planes] outputs: conv1, norm1+relu, conv2, downsample conv, its norm, block output */
  float *st1, *st2, *std_;               /* [2 planes] mean | invstd of the three norms (saved for backward) */
  void *conv_ws, *bn_ws;                 /* lgs_block_workspace_bytes / lgs_bn_workspace_bytes(n, planes) */
} lgs_block_fwd;
typedef struct lgs_block_bwd {
  lgs_kmap *km3, *km1;
  int dtype, relu_final, cin, planes, want_gin, x_row_stride;   /* x_row_stride: elements, 0 = cin (zero-copy cat slices) */
  int64_t n, dy_row_stride;              /* dy_row_stride: elements, 0 = planes */
  const void *dy;                        /* [n, planes] upstream gradient */
  const void *x, *o1, *y1, *o2, *y2, *od;               /* saved by the forward (y2 only when relu_final; od with km1) */
  const float *st1, *st2, *std_;
  const float *w1, *w2, *wd;
  void *pk1, *pk2, *pkd;                 /* packed DGRAD images (lgs_conv_pack_desc, op 1) or NULL */
  int pm1, pm2, pmd;
  const float *gamma1, *beta1, *gamma2, *beta2, *gammad, *betad;
  void *dx2, *dres, *dy1, *dx1, *dxd, *gind;            /* [n, planes] x5 scratch / results, gind [n, cin] (km1 only) */
  float *gw1, *gw2, *gwd;                /* weight gradients, fp32, parameter shapes (e.g. views of gradient-bucket slots) */
  float *dgamma1, *dbeta1, *dgamma2, *dbeta2, *dgammad, *dbetad;
  void *conv_ws, *bn_ws;
  /* weight gradients beside the dgrad / BatchNorm chain (me/modules.py conv_weight_grad does the same call by call): when
   * wgrad_stream is set, every lgs_conv_wgrad of the block is enqueued THERE, ordered after its operands by fork_event
   * (hipEvent_t, re-recorded on `stream` per weight gradient), uses wgrad_ws (its own lgs_conv_workspace_bytes(op 2) scratch)
   * and is followed by a record of ev_w1 / ev_w2 / ev_wd (hipEvent_t, may be NULL) on wgrad_stream -- what the consumer of
   * the gradient (bucketed all-reduce, optimizer) waits for.  NULL: the weight gradients run on `stream`. */
  void *wgrad_stream, *wgrad_ws, *fork_event, *ev_w1, *ev_w2, *ev_wd;
} lgs_block_bwd;                         /* grad_in: dres (no downsample) or gind (with), accumulated in place */
int64_t lgs_block_workspace_bytes(const lgs_kmap *km3, const lgs_kmap *km1, int cin, int planes, int dtype);
int lgs_block_forward(const lgs_block_fwd *args, void *stream);
int lgs_block_backward(const lgs_block_bwd *args, void *stream);

/* ---- CLIP text-anchor contraction (MFMA) ----------------------------------------------------
 * replaces ContrastiveLanguageLoss.feat_dist (cos) + feature_sim
 *   /root/reference/lib/losses/ContrastiveLanguageLoss.py:73-95,185-192
 *   /root/reference/lib/losses/utils.py:80-103
 * S[n, n_anchor] = normalize(F)[n,c] . normalize(T)[n_anchor,c]^T, float32 out.
 * inv_norm_f (float32 [n], may be NULL) receives 1/max(|f|,1e-12) for the backward. */
int lgs_clip_similarity(const void *feat, int64_t n, int c, const float *anchors, int n_anchor, float *sim,
                        float *inv_norm_f, int dtype, void *workspace, void *stream);
int64_t lgs_clip_workspace_bytes(int c, int n_anchor, int dtype);

/* ---- fused CLIP text-anchor loss (forward in ONE pass over the features, 4-sparse backward) --------
 * replaces ContrastiveLanguageLoss.forward's arithmetic (gather [N,1+K,C] + feat_dist bmm) AND feature_sim + argmax
 *   /root/reference/lib/losses/ContrastiveLanguageLoss.py:73-95 (feat_dist, cos), :185-192 (hinge inputs)
 *   /root/reference/lib/losses/utils.py:80-103 (feature_sim, cosine branch) + pl_RepresentationTrainer.py:237-238 (argmax)
 * lgs_clip_loss_forward: per row n (labels[n] == ignore_label, or outside [0, n_anchor): d_pos = d_neg = 0, :94)
 *   d_pos[n] = 1 - <f^_n, t^_labels[n]>,   d_neg[n] = 1 - mean_j <f^_n, t^_neg[n,j]>,   j < k_neg (1..7)
 *   pred[n]  = argmax_a <f^_n, t^_a>  (first maximum; may be NULL),   inv_norm_f[n] = 1 / max(|f_n|, 1e-12)
 *   anchors_n[n_anchor, c] (float32 out) = row-normalised anchors, kept for the backward
 *   sim[n, n_anchor] (float32) is written only when non-NULL (metrics / visualisation): the loss never needs it.
 *   n_anchor % 4 == 0, 4 <= n_anchor <= 224 (a wavefront owns all anchor columns of its rows).
 * lgs_clip_loss_backward: grad_feat[n, c] (dtype) from the upstream g_dpos[n], g_dneg[n] (float32, either may be NULL):
 *   gf = ( sum_j gs_j t^_j - (sum_j gs_j s_j) f^ ) / |f|  with  gs_pos = -g_dpos, gs_neg_j = -g_dneg / k_neg  (zero rows
 *   for ignored labels).  d_pos / d_neg / inv_norm_f / anchors_n are the forward's outputs. */
int64_t lgs_clip_loss_workspace_bytes(int c, int n_anchor, int dtype);
int lgs_clip_loss_forward(const void *feat, int64_t n, int c, const float *anchors, int n_anchor, const int64_t *labels,
                          const int64_t *neg, int k_neg, int64_t ignore_label, float *d_pos, float *d_neg, int64_t *pred,
                          float *inv_norm_f, float *anchors_n, float *sim, int dtype, void *workspace, void *stream);
int lgs_clip_loss_backward(const void *feat, int64_t n, int c, const float *anchors_n, int n_anchor, const int64_t *labels,
                           const int64_t *neg, int k_neg, int64_t ignore_label, const float *inv_norm_f, const float *d_pos,
                           const float *d_neg, const float *g_dpos, const float *g_dneg, void *grad_feat, int dtype,
                           void *stream);
/* lgs_clip_loss_backward_anchors: the same upstream gradient w.r.t. the NORMALISED anchors, for models that learn a projection of
 *   the text anchors (/root/reference/models/clip_models.py:192-200, Res16UNet34CR_Proj):
 *   grad_anchors_t[c][a8] (float32, a8 = n_anchor rounded up to 8; column a = d loss / d t^_a) = F^T G, G the 4-sparse matrix
 *   dL/dS (-g_dpos at the class, -g_dneg / k_neg at each negative, zero rows for ignored labels), as ONE launch of the weight-
 *   gradient kernels over the identity map (fixed summation order).  The caller applies d t^ -> d t (the anchor normalisation). */
int64_t lgs_clip_anchor_grad_workspace_bytes(int64_t n, int c, int n_anchor, int dtype);
int lgs_clip_loss_backward_anchors(const void *feat, int64_t n, int c, int n_anchor, const int64_t *labels, const int64_t *neg,
                                   int k_neg, int64_t ignore_label, const float *inv_norm_f, const float *g_dpos,
                                   const float *g_dneg, float *grad_anchors_t, int dtype, void *workspace, void *stream);

/* ---- fused SGD step on a flat parameter / gradient bucket -----------------------------------------
 * torch.optim.SGD's update rule as the reference configures it (/root/reference/lib/solvers.py: momentum 0.9,
 * dampening 0.1, weight_decay 1e-4) in ONE pass:  d = g + wd*p;  buf = first_step ? d : m*buf + (1-damp)*d;
 * p -= lr * mask * buf.   mask (may be NULL) zeroes the update of parameters that received no gradient this step. */
int lgs_sgd_step(float *params, const float *grads, float *momentum_buf, const float *mask, int64_t n, float lr,
                 float momentum, float dampening, float weight_decay, int first_step, void *stream);

/* ---- voxelisation on the device (the step in front of the hot path) ----------------
 * lgs_voxelize: points[n,3] float32 -> coords[n,4] int32 = (batch, floor(A * (x,y,z,1))), A = 3x4 row-major affine
 *   given as 12 HOST doubles (voxel scale / rotation / translation), evaluated in double like numpy's:
 *   /root/reference/lib/voxelizer.py:136-139 (homo_coords @ rigid_transformation.T[:, :3] -> np.floor) plus the batch
 *   column of ME.utils.sparse_collate (lib/transforms.py:421).
 * lgs_label_vote: label rule of ME.utils.sparse_quantize(..., labels, ignore_label) (lib/voxelizer.py:284): a voxel
 *   keeps its first point's label unless another of its points disagrees -> ignore_label.  unique_index / inverse are
 *   the outputs of lgs_manager_insert (device int64), labels / labels_out device int64.
 * Dedup (first occurrence wins, surviving indices ascending) is lgs_manager_insert itself. */
int lgs_voxelize(const float *points, int64_t n, const double *affine, int batch, int32_t *coords, void *stream);
int lgs_label_vote(const int64_t *labels, int64_t n, const int64_t *unique_index, const int64_t *inverse,
                   int64_t n_unique, int64_t ignore_label, int64_t *labels_out, void *stream);

/* ---- PointGroup clustering (validation-time only) --------------------------------
 * replaces PG_OP.ballquery_batch_p + PG_OP.bfs_cluster
 *   /root/reference/downstream/insseg/lib/bfs/ops/src/bfs_cluster_kernel.cu:16-61, bfs_cluster.cpp:54-125,
 *   as called by downstream/insseg/lib/bfs/bfs.py:124-150.
 * Connected components of {(i,j): |p_i - p_j|^2 < radius^2 (float32, reference operation order), same batch, same
 * semantic label} through a radius-sized cell grid + lock-free union-find; component[i] = smallest point index of i's
 * component (= the point the reference's BFS starts it from, so clusters sorted by it come in the reference's order)
 * or -1 if the component has fewer than `threshold` points; *n_clusters (host) = kept components.  The reference's
 * per-point cap of 1000 neighbours / meanActive buffer is not reproduced (no neighbour lists are materialised).
 * Synchronises `stream` once.  batch_idx may be NULL (single scene). */
int64_t lgs_cluster_workspace_bytes(int64_t n);
int lgs_cluster(const float *xyz, const int32_t *batch_idx, const int32_t *semantic_label, int64_t n, float radius,
                int threshold, int32_t *component, int32_t *n_clusters, void *workspace, void *stream);

/* ---- fused softmax cross-entropy ----------------------------------------------------------------
 * replaces nn.CrossEntropyLoss(ignore_index=-1) on the [N,200] logits of the fine-tune step
 *   /root/reference/lib/train_test/pl_BaselineTrainer.py:94-99,350
 * One pass: loss_rows[n] (float32, 0 for ignored rows) and dlogits[n,c] = (softmax - onehot) * (*scale)
 * (same dtype as logits, zeros for ignored rows; rows whose label is outside [0, c) are treated as ignored).  Any class
 * count up to 512 (fp32) / 1024 (bf16) is accepted; counts that are not a multiple of the 16-byte width (20 ScanNet
 * classes in bf16) take element-wise accesses.  `scale` is a DEVICE float (e.g. 1 / #valid rows, times
 * the upstream gradient), so the mean reduction needs no host sync.  Either output may be NULL: the host
 * wrapper asks for the loss in the forward pass and for the gradient in the backward pass. */
int lgs_ce_forward_backward(const void *logits, int64_t n, int c, const int64_t *labels, int64_t ignore_index,
                            const float *scale, float *loss_rows, void *dlogits, int dtype, void *stream);
/* The same pass with a per-row gradient factor (ABI 13): dlogits[n,:] = (softmax - onehot) * (*scale) * row_scale[n].
 * This is the backward of nn.CrossEntropyLoss(reduction='none') -- what the fine-tune step runs when
 * --balanced_category_sampling True (/root/reference/scripts/train_models.sh:37, pl_BaselineTrainer.py:94,350-356):
 * loss_rows IS the reduction='none' output, row_scale the upstream gradient of sample_categories_for_balancing's masked mean
 * (lib/losses/utils.py:74-77: mask / N).  row_scale may be NULL (= lgs_ce_forward_backward). */
int lgs_ce_forward_backward_rows(const void *logits, int64_t n, int c, const int64_t *labels, int64_t ignore_index,
                                 const float *scale, const float *row_scale, float *loss_rows, void *dlogits, int dtype,
                                 void *stream);
/* head / common / tail statistics of per-point losses (ABI 13): what the trainer's three meters take from
 * sample_categories_for_balancing (/root/reference/lib/losses/utils.py:69-72, pl_BaselineTrainer.py:353-355) without its
 * three boolean-index gathers (= three host syncs).  group_of_class[n_classes] in {0,1,2} (anything else: not counted); rows whose
 * label is ignore_index or outside [0, n_classes) are skipped.  partial[partial_rows][6] (1 <= partial_rows <= 1024, overwritten):
 * per workgroup (sum, count) x 3 groups -- the caller adds the rows up (deterministic; no float atomics). */
int lgs_split_stats(const float *loss_rows, const int64_t *labels, int64_t n, const int32_t *group_of_class, int n_classes,
                    int64_t ignore_index, float *partial, int partial_rows, void *stream);
/* number of rows the loss above counts (label != ignore_index and inside [0, c)) -> *count (DEVICE int32, overwritten): the
 * denominator of the mean reduction (pl_BaselineTrainer.py:350, nn.CrossEntropyLoss(ignore_index) 'mean') without a host
 * sync and without a chain of elementwise / reduction launches over the label tensor. */
int lgs_ce_count_valid(const int64_t *labels, int64_t n, int c, int64_t ignore_index, int32_t *count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LGS_ENGINE_H */
