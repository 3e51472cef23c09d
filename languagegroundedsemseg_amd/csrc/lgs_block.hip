// lgs_block.hip -- one C-ABI call per residual block and direction.
//
// A BasicBlock (/root/reference/models/modules/resnet_block.py:41-57: conv3-norm-relu-conv3-norm-(+residual)-relu, with the
// optional 1x1 conv + norm downsample branch of /root/reference/models/resnet.py:93-103) is a FIXED sequence of engine calls:
// forward 2-3 lgs_conv_forward + 2-3 lgs_bn_forward, backward 2-3 each of lgs_bn_backward / lgs_conv_wgrad / lgs_conv_dgrad.
// At one ~150 k-voxel scene per step the training step is bound by the host enqueueing those ~250 calls (~40 us each through
// Python + ctypes: host 10.2 ms against 9.0 ms of GPU work), not by the GPU.  These two entry points
// issue exactly the same launches, in the same order, with the same arguments as the call-by-call path (bit-identical results:
// tests/test_gpu_parity_r4.py::test_c_side_block_equals_the_call_by_call_block) from ONE host call each.  Everything runs on
// the caller's stream, except the weight gradients when the caller names a side stream for them (lgs_block_bwd.wgrad_stream).
#include "lgs_common.h"

extern "C" {

int64_t lgs_block_workspace_bytes(const lgs_kmap *km3, const lgs_kmap *km1, int cin, int planes, int dtype) {
  if (!km3) return -1;
  int64_t b = 0;
  auto up = [&](int64_t v) { if (v > b) b = v; };
  for (int op = 0; op < 3; ++op) {
    up(lgs_conv_workspace_bytes(km3, cin, planes, dtype, op));
    up(lgs_conv_workspace_bytes(km3, planes, planes, dtype, op));
    if (km1) up(lgs_conv_workspace_bytes(km1, cin, planes, dtype, op));
  }
  return b;
}

int lgs_block_forward(const lgs_block_fwd *a, void *stream) {
  LGS_REQUIRE(a && a->km3 && a->x && a->w1 && a->w2 && a->o1 && a->y1 && a->o2 && a->y2 && a->conv_ws && a->bn_ws,
              "lgs_block_forward: null argument");
  LGS_REQUIRE((a->km1 != nullptr) == (a->wd != nullptr), "lgs_block_forward: downsample map and weight go together");
  const int dt = a->dtype, c = a->planes;
  int rc;
  if ((rc = lgs_conv_forward(a->km3, 0, a->x, a->cin, a->w1, c, nullptr, a->o1, dt, a->conv_ws, nullptr, nullptr, a->pk1, a->pm1, 0, stream))) return rc;
  if ((rc = lgs_bn_forward(a->o1, a->n, c, a->n1.gamma, a->n1.beta, a->n1.eps, a->n1.momentum, a->n1.running_mean, a->n1.running_var,
                           a->n1.num_batches_tracked, nullptr, 1, a->y1, a->st1, dt, a->bn_ws, nullptr, 0, nullptr, 0, stream))) return rc;
  if ((rc = lgs_conv_forward(a->km3, 0, a->y1, c, a->w2, c, nullptr, a->o2, dt, a->conv_ws, nullptr, nullptr, a->pk2, a->pm2, 0, stream))) return rc;
  const void *res = a->x;
  if (a->km1) {
    LGS_REQUIRE(a->od && a->res && a->std_, "lgs_block_forward: downsample branch outputs missing");
    if ((rc = lgs_conv_forward(a->km1, 0, a->x, a->cin, a->wd, c, nullptr, a->od, dt, a->conv_ws, nullptr, nullptr, a->pkd, a->pmd, 0, stream))) return rc;
    if ((rc = lgs_bn_forward(a->od, a->n, c, a->nd.gamma, a->nd.beta, a->nd.eps, a->nd.momentum, a->nd.running_mean, a->nd.running_var,
                             a->nd.num_batches_tracked, nullptr, 0, a->res, a->std_, dt, a->bn_ws, nullptr, 0, nullptr, 0, stream))) return rc;
    res = a->res;
  } else {
    LGS_REQUIRE(a->cin == c, "lgs_block_forward: a block without a downsample branch keeps its width");
  }
  return lgs_bn_forward(a->o2, a->n, c, a->n2.gamma, a->n2.beta, a->n2.eps, a->n2.momentum, a->n2.running_mean, a->n2.running_var,
                        a->n2.num_batches_tracked, res, a->relu_final, a->y2, a->st2, dt, a->bn_ws, nullptr, 0, nullptr, 0, stream);
}

int lgs_block_backward(const lgs_block_bwd *a, void *stream) {
  LGS_REQUIRE(a && a->km3 && a->x && a->dy && a->o1 && a->y1 && a->o2 && a->st1 && a->st2 && a->w1 && a->w2 && a->dx2 && a->dres && a->dy1 &&
                  a->dx1 && a->gw1 && a->gw2 && a->conv_ws && a->bn_ws,
              "lgs_block_backward: null argument");
  LGS_REQUIRE(!a->relu_final || a->y2, "lgs_block_backward: the ReLU mask of the block output needs y2");
  const int dt = a->dtype, c = a->planes;
  int rc;
  // one weight gradient: on the caller's stream, or on the side stream behind a fork event (its operands are complete on `stream`)
  auto wgrad = [&](lgs_kmap *km, const void *in, int cin, const void *gout, float *gw, int in_ld, void *ev) -> int {
    if (!a->wgrad_stream) return lgs_conv_wgrad(km, 0, in, cin, gout, c, gw, dt, a->conv_ws, in_ld, stream);
    LGS_REQUIRE(a->fork_event && a->wgrad_ws, "lgs_block_backward: a side stream needs fork_event and wgrad_ws");
    LGS_HIP(hipEventRecord((hipEvent_t)a->fork_event, (hipStream_t)stream));
    LGS_HIP(hipStreamWaitEvent((hipStream_t)a->wgrad_stream, (hipEvent_t)a->fork_event, 0));
    const int r = lgs_conv_wgrad(km, 0, in, cin, gout, c, gw, dt, a->wgrad_ws, in_ld, a->wgrad_stream);
    if (r) return r;
    if (ev) LGS_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)a->wgrad_stream));
    return 0;
  };
  // norm2 (+ residual) (+ ReLU): mask from the saved output when there is a ReLU (a residual was added)
  if ((rc = lgs_bn_backward(a->o2, a->relu_final ? a->y2 : nullptr, a->dy, a->dy_row_stride, a->n, c, a->gamma2, a->beta2, a->st2,
                            a->relu_final ? 1 : 0, a->dx2, a->dres, a->dgamma2, a->dbeta2, dt, a->bn_ws, 0, stream))) return rc;
  if ((rc = wgrad(a->km3, a->y1, c, a->dx2, a->gw2, 0, a->ev_w2))) return rc;
  if ((rc = lgs_conv_dgrad(a->km3, 0, a->dx2, c, a->w2, c, a->dy1, dt, a->conv_ws, a->pk2, a->pm2, stream))) return rc;
  // norm1 + ReLU: mask recomputed from its input
  if ((rc = lgs_bn_backward(a->o1, nullptr, a->dy1, 0, a->n, c, a->gamma1, a->beta1, a->st1, 2, a->dx1, nullptr, a->dgamma1, a->dbeta1, dt,
                            a->bn_ws, 0, stream))) return rc;
  // (A/B knob BLOCK_WGRAD_LATE: the first convolution's weight gradient is forked AFTER the block's last dgrad instead of beside it)
  const bool late = lgs::tune(lgs::T_BLOCK_WGRAD_LATE) != 0 && a->wgrad_stream && a->want_gin;
  if (!late && (rc = wgrad(a->km3, a->x, a->cin, a->dx1, a->gw1, a->x_row_stride, a->ev_w1))) return rc;
  void *acc = a->dres;       // the residual branch's gradient w.r.t. x
  if (a->km1) {
    LGS_REQUIRE(a->od && a->std_ && a->wd && a->dxd && a->gwd && a->gind, "lgs_block_backward: downsample branch tensors missing");
    if ((rc = lgs_bn_backward(a->od, nullptr, a->dres, 0, a->n, c, a->gammad, a->betad, a->std_, 0, a->dxd, nullptr, a->dgammad, a->dbetad, dt,
                              a->bn_ws, 0, stream))) return rc;
    if ((rc = wgrad(a->km1, a->x, a->cin, a->dxd, a->gwd, a->x_row_stride, a->ev_wd))) return rc;
    if (a->want_gin) {
      if ((rc = lgs_conv_dgrad(a->km1, 0, a->dxd, c, a->wd, a->cin, a->gind, dt, a->conv_ws, a->pkd, a->pmd, stream))) return rc;
      acc = a->gind;
    }
  }
  if (!a->want_gin) return 0;
  // grad_in = dgrad1(dx1) + (gradient of the residual branch): in the epilogue where the launch shape has an accumulating one
  LGS_REQUIRE(lgs_conv_dgrad_can_accumulate(a->km3, 0, a->cin, c, dt), "lgs_block_backward: this shape needs the call-by-call path (no accumulating dgrad)");
  if ((rc = lgs_conv_dgrad_accumulate(a->km3, 0, a->dx1, c, a->w1, a->cin, acc, dt, a->conv_ws, a->pk1, a->pm1, stream))) return rc;
  if (late) return wgrad(a->km3, a->x, a->cin, a->dx1, a->gw1, a->x_row_stride, a->ev_w1);
  return 0;
}

}  // extern "C"
