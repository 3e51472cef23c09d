"""Round-6 GPU parity: the fine-tune step as scripts/train_models.sh runs it.

  /root/reference/scripts/train_models.sh:37                  --balanced_category_sampling True
  /root/reference/lib/train_test/pl_BaselineTrainer.py:94     reduction = 'none' if balanced_category_sampling
  /root/reference/lib/train_test/pl_BaselineTrainer.py:350-356 loss = criterion(logits, target); sample_categories_for_balancing
  /root/reference/lib/losses/utils.py:13-77                   masked mean over ALL points

* lgs_ce_forward_backward_rows (per-row loss, per-row upstream gradient) vs nn.CrossEntropyLoss(reduction='none') + autograd;
* the whole balanced loss (per-point CE -> balancing -> backward) under torch.cuda.set_sync_debug_mode("error"): no host sync;
* one BASELINE-shaped scene (150k voxels @2cm, 200 classes, fp32) through Res16UNet34C forward + loss + backward against the
  oracle (pl_BaselineTrainer.py:300-305, models/res16unet.py:196-270): logits <= 1e-3, loss <= 1e-4, gradient rel-L2 <= 1e-2."""
import os

import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from helpers import Cfg, deterministic_init
from languagegroundedsemseg_amd.models import load_model
from languagegroundedsemseg_amd.synthetic import make_batch, text_anchors
from oracle.backend import OracleBackend
from test_gpu_parity_r2 import grad_report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


@pytest.mark.parity("nn.CrossEntropyLoss(reduction='none') + autograd in torch float32")
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("classes", [200, 20])
def test_per_point_cross_entropy_and_its_row_scaled_gradient(classes, dtype, tol):
    from languagegroundedsemseg_amd.losses import fused_cross_entropy
    torch.manual_seed(classes)
    n = 5003
    x = (torch.randn(n, classes) * 3).to(dtype).float()
    lab = torch.randint(-1, classes, (n,))
    lab[7] = classes + 3                                   # outside [0, C): an ignored row, as on every engine loss path
    g = torch.rand(n) * 2 - 0.5
    xr = x.clone().requires_grad_(True)
    lab_t = torch.where((lab >= 0) & (lab < classes), lab, torch.full_like(lab, -1))
    ref = torch.nn.functional.cross_entropy(xr, lab_t, ignore_index=-1, reduction="none")
    (ref * g).sum().backward()
    xh = x.to(DEV).to(dtype).requires_grad_(True)
    rows = fused_cross_entropy(xh, lab.to(DEV), ignore_index=-1, reduction="none")
    assert rows.shape == (n,) and rows.dtype == torch.float32
    (rows * g.to(DEV)).sum().backward()
    assert float((rows.cpu() - ref.detach()).abs().max()) <= tol * 10
    assert bool((rows.cpu()[lab_t == -1] == 0).all())
    assert float((xh.grad.float().cpu() - xr.grad).abs().max()) <= tol
    assert bool((xh.grad.float().cpu()[lab_t == -1] == 0).all())


def test_balanced_fine_tune_loss_runs_without_a_host_sync():
    """per-point CE -> sample_categories_for_balancing(split='stats') -> backward with torch's sync debugger set to raise; the
    value equals the split='tensors' (reference-shaped) result drawn from the same generator state"""
    from languagegroundedsemseg_amd.losses import fused_cross_entropy, sample_categories_for_balancing
    torch.manual_seed(3)
    n, L = 40000, 200
    foc = torch.zeros(L, 3, dtype=torch.bool)
    foc[:66, 0], foc[66:134, 1], foc[134:, 2] = True, True, True
    foc = foc.to(DEV)
    logits = torch.randn(n, L, device=DEV).to(torch.bfloat16).requires_grad_(True)
    lab = torch.randint(-1, L, (n,), device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(11)
    state = gen.get_state()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for hr, cr in ((-1.0, -1.0), (0.5, 0.25)):
            rows = fused_cross_entropy(logits, lab, ignore_index=-1, reduction="none")
            loss, stats, items = sample_categories_for_balancing(rows, lab, foc, hr, cr, ignore_label=-1, generator=gen, split="stats")
            loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    g_stats = logits.grad.clone()
    logits.grad = None
    gen.set_state(state)
    rows = fused_cross_entropy(logits, lab, ignore_index=-1, reduction="none")
    sample_categories_for_balancing(rows, lab, foc, -1.0, -1.0, ignore_label=-1, generator=gen)
    ref_loss, (head, common, tail), ref_items = sample_categories_for_balancing(rows, lab, foc, 0.5, 0.25, ignore_label=-1, generator=gen)
    assert float((loss - ref_loss).abs()) <= 1e-6
    for i, t in enumerate((head, common, tail)):
        assert int(stats[i, 1]) == t.numel() and abs(float(stats[i, 0]) - float(t.double().sum())) <= 1e-3 * t.numel() ** 0.5 + 1e-3
    assert torch.equal(items[lab != -1], ref_items)
    # gradient: kept rows carry (softmax - onehot) / N, dropped and ignored rows zero; the two accumulated backward passes add up
    sm = torch.softmax(logits.detach().float(), 1)
    sm[torch.arange(n, device=DEV), lab.clamp_min(0)] -= 1.0
    valid = (lab != -1).float()[:, None]
    gen.set_state(state)
    rows = fused_cross_entropy(logits, lab, ignore_index=-1, reduction="none")
    la, _, _ = sample_categories_for_balancing(rows, lab, foc, -1.0, -1.0, ignore_label=-1, generator=gen, split="stats")
    lb, _, _ = sample_categories_for_balancing(rows, lab, foc, 0.5, 0.25, ignore_label=-1, generator=gen, split="stats")
    (la + lb).backward()
    assert float((logits.grad.float() - g_stats.float()).abs().max()) <= 2e-7      # bf16 rounding of ~1e-5-sized entries
    assert float((g_stats.float().abs().sum(1) * (1 - valid[:, 0])).max()) == 0.0
    full = sm * valid / n                                   # the ratios -1 pass alone
    kept_b = g_stats.float() - full.to(torch.bfloat16).float()
    frac = float((kept_b.abs().sum(1) > 0).float().sum() / valid.sum())
    assert 0.45 <= frac <= 0.75                             # 66 head classes at 0.5, 68 common at 0.25, 66 tail at 1.0 -> ~0.58


def _step_34c(coords, feats, labels, device, n_classes):
    m = deterministic_init(load_model("Res16UNet34C")(3, n_classes, Cfg()), 42).to(device).train()
    x = ME.SparseTensor(torch.from_numpy(feats).to(device), torch.from_numpy(coords).to(device))
    logits, _ = m(x)
    loss = torch.nn.functional.cross_entropy(logits.F.float(), torch.from_numpy(labels).to(device), ignore_index=-1)
    loss.backward()
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    return logits.F.detach().float().cpu().numpy(), float(loss), grads


def test_res16unet34c_fp32_at_the_baseline_shape_150k_voxels_200_classes_vs_oracle():
    """BASELINE configs[1]'s unit: ONE ~150k-voxel 2 cm scene, 200 classes, fp32, forward + CE + backward, against the oracle's
    BLAS gather-GEMM-scatter backend (what `cpu_baseline` times): north_star's "logits within 1e-3 fp32" at the shape it is quoted on"""
    coords, feats, labels = make_batch([7], voxel=0.02, n_target=150000)
    feats = feats / 255.0 - 0.5                             # pl_BaselineTrainer.py:298-299
    prev = ME.set_backend(OracleBackend("torch"))
    try:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        lo, so, go = _step_34c(coords, feats, labels, "cpu", 200)
    finally:
        ME.set_backend(prev)
    lh, sh, gh = _step_34c(coords, feats, labels, DEV, 200)
    assert lh.shape == lo.shape == (coords.shape[0], 200) and coords.shape[0] > 140000
    err = float(np.abs(lh - lo).max())
    print("150k-voxel 34C fp32: max |logit - oracle| = %.3g, loss %.6f vs %.6f" % (err, sh, so))
    assert err <= 1e-3
    assert abs(sh - so) <= 1e-4
    assert set(gh) == set(go)
    _, tot = grad_report(gh, go, "34C fp32 150k voxels / 200 classes vs oracle")
    worst = max((rel_l2(gh[k], go[k]), k) for k in go)
    # over all parameters <= 1e-2 (measured ~3e-3); a single small tensor of the coarsest level may sit a little above that: ReLU
    # gates of ~600 level-4 voxels flipping at fp32 round-off (block4.3.norm1.bn.bias: 1.3e-2), cf. tests/test_gpu_reference_calls.py
    assert tot <= 1e-2 and worst[0] <= 5e-2, (tot, worst)


def test_res16unet34d_clip_loss_fp32_at_150k_voxels_vs_oracle():
    """BASELINE configs[2] at the same scene size: Res16UNet34D (512-d features) + the text-anchor contrastive loss with explicit
    negatives (pl_RepresentationTrainer.py:183-216), fp32, forward + loss + backward vs the oracle"""
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
    coords, feats, labels = make_batch([8], voxel=0.02, n_target=150000)
    feats = feats / 255.0 - 0.5
    anchors = text_anchors(200, 512)
    rng = np.random.RandomState(0)
    neg = rng.randint(0, 199, (coords.shape[0], 3))
    neg = neg + (neg >= np.maximum(labels, 0)[:, None])

    def step(device):
        m = deterministic_init(load_model("Res16UNet34D")(3, 200, Cfg()), 42).to(device).train()
        m.representation_only(True)
        x = ME.SparseTensor(torch.from_numpy(feats).to(device), torch.from_numpy(coords).to(device))
        out = m(x)
        crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
        loss = crit(out.F, torch.from_numpy(labels).to(device), torch.from_numpy(anchors).to(device),
                    neg_indices=torch.from_numpy(neg).to(device))[0]
        loss.backward()
        grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
        return out.F.detach().float().cpu().numpy(), float(loss), grads

    prev = ME.set_backend(OracleBackend("torch"))
    try:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        fo, so, go = step("cpu")
    finally:
        ME.set_backend(prev)
    fh, sh, gh = step(DEV)
    scale = float(np.abs(fo).max())
    err = float(np.abs(fh - fo).max())
    print("150k-voxel 34D fp32: max |feature - oracle| = %.3g (features up to %.3g), loss %.6f vs %.6f" % (err, scale, sh, so))
    assert err <= 1e-3 * max(1.0, scale)
    assert abs(sh - so) <= 1e-4
    assert set(gh) == set(go)
    _, tot = grad_report(gh, go, "34D + CLIP loss fp32 150k voxels vs oracle")
    worst = max((rel_l2(gh[k], go[k]), k) for k in go)
    assert tot <= 1e-2 and worst[0] <= 5e-2, (tot, worst)


def test_toggling_fp32_split_on_a_live_kernel_map_repacks_for_the_new_layout():
    """advisor (round 5): the pack descriptor of a launch shape is cached per kernel map; FP32_SPLIT changes the packed layout (three
    bf16 planes = 6 bytes per element instead of 4), so a descriptor cached across engine.tuning(...) fed a buffer of the old size
    / a "valid" image of the old layout to the new kernel.  Same SparseTensor (same manager, same map) through both modes and
    back: each mode equals the oracle, and the third call equals the first bit for bit."""
    from languagegroundedsemseg_amd import engine
    coords, feats, _ = make_batch([3], voxel=0.05, n_target=20000)
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(32, 64, kernel_size=3, dimension=3).to(DEV)
    f = torch.randn(coords.shape[0], 32)
    prev = ME.set_backend(OracleBackend("torch"))
    try:
        ref = conv.cpu()(ME.SparseTensor(f, torch.from_numpy(coords))).F.detach()
    finally:
        ME.set_backend(prev)
    conv = conv.to(DEV)
    x = ME.SparseTensor(f.to(DEV), torch.from_numpy(coords).to(DEV))
    with torch.no_grad():
        assert engine.tuning_get("FP32_SPLIT") == 1
        y1 = conv(x).F.clone()
        with engine.tuning(FP32_SPLIT=0):
            y0 = conv(x).F.clone()
        y2 = conv(x).F.clone()
    scale = float(ref.abs().max())
    assert float((y1.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, scale)
    assert float((y0.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, scale)
    assert torch.equal(y1, y2)


@pytest.mark.parity("torch semantics of zero-row tensors")
def test_an_empty_batch_flows_through_the_operator_surface():
    """a collate that dropped every scene (lib/transforms.py:402-412 drops whole scenes over the point limit) hands the model a
    [0, 4] coordinate tensor: insert, strided maps, 3^3 / 2^3 s2 / transposed / 1x1 convolutions, cat and the losses give zero-row
    results (BatchNorm in eval mode: batch statistics of nothing are undefined in torch as well), backward runs, nothing is launched
    with an empty grid"""
    from languagegroundedsemseg_amd.losses import fused_cross_entropy
    x = ME.SparseTensor(torch.zeros(0, 3, device=DEV), torch.zeros(0, 4, dtype=torch.int32, device=DEV))
    assert x.F.shape == (0, 3) and x.C.shape == (0, 4)
    c0 = ME.MinkowskiConvolution(3, 32, kernel_size=3, dimension=3).to(DEV)
    dn = ME.MinkowskiConvolution(32, 32, kernel_size=2, stride=2, dimension=3).to(DEV)
    up = ME.MinkowskiConvolutionTranspose(32, 32, kernel_size=2, stride=2, dimension=3).to(DEV)
    fin = ME.MinkowskiConvolution(64, 20, kernel_size=1, bias=True, dimension=3).to(DEV)
    bn = ME.MinkowskiBatchNorm(32).to(DEV).eval()
    with pytest.raises(ValueError, match="more than 1 value per channel"):      # nn.BatchNorm1d's own answer in training mode
        ME.MinkowskiBatchNorm(32).to(DEV).train()(c0(x)).F
    a = bn(c0(x))
    b = up(dn(a))
    out = fin(ME.cat(b, a))
    assert out.F.shape == (0, 20) and out.C.shape == (0, 4)
    rows = fused_cross_entropy(out.F, torch.zeros(0, dtype=torch.int64, device=DEV), -1, reduction="none")
    assert rows.shape == (0,)
    loss = fused_cross_entropy(out.F, torch.zeros(0, dtype=torch.int64, device=DEV), -1)
    assert float(loss) == 0.0
    (loss + rows.sum()).backward()
    for m in (c0, dn, up, fin):
        assert m.kernel.grad is not None and float(m.kernel.grad.abs().sum()) == 0.0


@pytest.mark.parity("the oracle's 1x1 convolution")
def test_streaming_pointwise_conv_equals_the_gather_kernel_and_reads_column_slices():
    """k_pointwise (1x1 layers of the big maps, lgs_pointwise.hip) against k_conv_gather's identity-map tiles (POINTWISE=0) on the
    same tensors -- forward with bias and dgrad, a row count that is not a multiple of 32, an input that is a column slice of a
    wider buffer (the zero-copy cat hands such slices to the downsample branch) -- and both against the oracle"""
    from languagegroundedsemseg_amd import engine
    coords, _, _ = make_batch([4], voxel=0.02, n_target=80000)
    n = coords.shape[0]
    assert n >= 66000 and n % 32 != 0
    torch.manual_seed(1)
    x = ME.SparseTensor(torch.zeros(n, 3, device=DEV), torch.from_numpy(coords).to(DEV))
    km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 1)
    wide = torch.randn(n, 160, device=DEV).bfloat16()
    _pointwise_cases(km, n, wide, torch.bfloat16, 1e-2)


@pytest.mark.parity("the oracle's 1x1 convolution")
def test_streaming_pointwise_conv_fp32_on_the_exact_mfma():
    """k_pointwise_f32 (v_mfma_f32_32x32x2_f32: every product and sum in fp32) against the identity-map tiles of k_conv_gather and
    against a float64 product: 2e-5 of the result's scale (the per-op bar of the fp32 parity path); same shapes, ragged rows and
    column slices as the bf16 test"""
    coords, _, _ = make_batch([4], voxel=0.02, n_target=80000)
    n = coords.shape[0]
    torch.manual_seed(2)
    x = ME.SparseTensor(torch.zeros(n, 3, device=DEV), torch.from_numpy(coords).to(DEV))
    km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 1)
    _pointwise_cases(km, n, torch.randn(n, 160, device=DEV), torch.float32, 2e-5)


def _pointwise_cases(km, n, wide, dtype, tol):
    from languagegroundedsemseg_amd import engine
    for cin, cout, sl in ((96, 200, None), (128, 96, slice(32, 160)), (96, 128, slice(0, 96))):
        f = wide[:, :cin].contiguous() if sl is None else wide[:, sl]
        w = torch.randn(1, cin, cout, device=DEV) * 0.1
        b = torch.randn(1, cout, device=DEV)
        g = torch.randn(n, cout, device=DEV).to(dtype)
        engine.dispatch_counts(reset=True)
        with engine.tuning(POINTWISE=2):         # every shape the kernel serves (production keeps the narrow ones on k_conv_gather)
            y1, d1 = km.conv_forward(f, w, b, False), km.conv_dgrad(g, w, False)
        sites = engine.dispatch_counts(reset=True)
        assert sum(v for k, v in sites.items() if k.startswith("k_pointwise")) == 2, sites
        with engine.tuning(POINTWISE=0):
            y0, d0 = km.conv_forward(f.contiguous(), w, b, False), km.conv_dgrad(g, w, False)
            assert not any(k.startswith("k_pointwise") for k in engine.dispatch_counts(reset=True))
        wq = w[0].to(dtype).double().cpu() if dtype == torch.bfloat16 else w[0].double().cpu()   # (bf16: the kernels multiply bf16 weights)
        ref = f.double().cpu() @ wq + b.double().cpu()
        dref = g.double().cpu() @ wq.t()
        for got, old, want in ((y1, y0, ref), (d1, d0, dref)):
            scale = float(want.abs().max())
            assert float((got.double().cpu() - want).abs().max()) <= tol * scale            # bf16: one rounding of the result
            assert float((got.float() - old.float()).abs().max()) <= tol * scale            # (the two kernels round the same sums)
