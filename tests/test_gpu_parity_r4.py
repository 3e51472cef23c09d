"""Round-4 parity additions (GPU):
  * k_wgrad_wide per op at the size production selects it (>= 200 k positions): 3^3 256 -> 256 and 512 -> 256 weight gradients
    vs the oracle, plus the strided-input (column slice of a concat buffer) reader;
  * train -> eval -> train -> eval BatchNorm: the cached eval-mode statistics must follow the running statistics the engine
    writes through raw pointers (round-3 advisor finding, high);
  * the whole-block fast path declines blocks whose convolutions are not the 3^3 / 3^3 / 1x1 stride-1 shape it hard-codes.
Reference: /root/reference/models/clip_models.py:205-215, models/modules/resnet_block.py:41-57, models/modules/common.py:17-19."""
import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from helpers import Cfg, deterministic_init
from oracle.backend import OracleBackend
from test_gpu_engine import rel_err, run_both

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _wide_hits(engine):
    return {k: v for k, v in engine.dispatch_counts().items() if k.split()[0] in ("k_wgrad_wide", "k_ww_count", "k_ww_scan", "k_ww_write",
                                                                                  "k_wgrad_wide_reduce")}


@pytest.mark.parametrize("cin,cout", [(256, 256), (512, 256)])
def test_wide_weight_gradient_at_production_size_matches_oracle(cin, cout):
    """two 2 cm scenes = a map of >= 210 k positions: above the WW_MIN_ROWS gate (200 k), i.e. the launch the 8-scene
    benchmark batch of BASELINE configs[2] takes for every >= 256 x 256-channel 3^3 layer.  bf16 forward / dgrad / wgrad vs
    the fp32 oracle on bf16-rounded inputs, 2e-2 relative (max-norm) like every per-op bf16 test."""
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([11, 12], voxel=0.02, n_target=112000)
    assert coords.shape[0] >= 210000, coords.shape
    assert engine.tuning_get("WW_MIN_ROWS") == 200000
    feats = torch.from_numpy(np.random.default_rng(5).standard_normal((coords.shape[0], cin)).astype(np.float32)).bfloat16().float().numpy()
    engine.dispatch_counts(reset=True)
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3)], coords, feats, dtype=torch.bfloat16,
        oracle_impl="torch")
    hits = _wide_hits(engine)
    assert len(hits) == 5 and all(v == 1 for v in hits.values()), hits      # count, scan, write, GEMM, reduce: one launch each
    assert rel_err(h_out, o_out) < 2e-2
    for n, a, b in zip(["dgrad", "wgrad"], h_g, o_g):
        assert rel_err(a, b) < 2e-2, n
    # and the sharper measure for a reduction over ~3 M pairs per offset: relative L2 over the whole [27, cin, cout] gradient
    a, b = h_g[1].astype(np.float64), o_g[1].astype(np.float64)
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 3e-3


def test_wide_weight_gradient_reads_a_column_slice_in_place():
    """the gathered operand as a column slice of a wider row-major buffer (a zero-copy ME.cat half: 512 of 544 columns):
    bit-identical to the contiguous copy, on k_wgrad_wide (in_row_stride)"""
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([13, 14], voxel=0.02, n_target=112000)
    n = coords.shape[0]
    x = ME.SparseTensor(torch.zeros(n, 3, device=DEV), torch.from_numpy(coords).to(DEV))
    km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
    g = torch.Generator(device=DEV).manual_seed(3)
    big = torch.randn(n, 544, device=DEV, generator=g).bfloat16()
    gout = torch.randn(n, 256, device=DEV, generator=g).bfloat16()
    sl = big[:, :512]
    assert not sl.is_contiguous()
    assert engine.lib().lgs_conv_wgrad_supports_stride(km.h, 0, 512, 256, engine.LGS_BF16, 544) == 1
    engine.dispatch_counts(reset=True)
    a = km.conv_wgrad(sl, gout, False)
    b = km.conv_wgrad(sl.contiguous(), gout, False)
    torch.cuda.synchronize()
    hits = _wide_hits(engine)
    assert hits and all(v == 2 for v in hits.values()), hits
    assert torch.equal(a, b)
    # a second, offset slice (columns 32 .. 543) as well
    sl2 = big[:, 32:544]
    assert torch.equal(km.conv_wgrad(sl2, gout, False), km.conv_wgrad(sl2.contiguous(), gout, False))


@pytest.mark.parametrize("cin,cout,bias,dtype,tol", [(96, 200, True, torch.float32, 2e-5), (96, 200, True, torch.bfloat16, 2e-2),
                                                      (128, 96, False, torch.bfloat16, 2e-2), (96, 160, True, torch.bfloat16, 2e-2),
                                                      (192, 128, False, torch.float32, 2e-5), (192, 128, False, torch.bfloat16, 2e-2),
                                                      (512, 256, False, torch.bfloat16, 2e-2), (256, 512, False, torch.bfloat16, 2e-2)])
def test_pointwise_conv_tiles_of_the_big_maps(cin, cout, bias, dtype, tol):
    """1x1 convolutions on maps of >= 65536 positions take tile configurations no small-map test reaches (found by
    tests/test_gpu_dispatch_coverage.py): the 7-block classifier tile (`final`, 96 -> 200 + bias, res16unet.py:193) in fp32 and
    bf16, and the 8-wave 256-channel tile of the representation model's 1x1 downsample branches (clip_models.py:205-215) in both
    directions (forward 512 -> 256 is its dgrad's 256 -> 512 shape and vice versa).  Round 6: the bf16 shapes with <= 224 output
    channels and >= 5 output blocks (or > 128 reduction channels) run on k_pointwise (lgs_pointwise.hip): 96 -> 200 forward =
    instance <7, 8>, its dgrad 200 -> 96 = <3, 16>; 96 -> 160 = five output blocks on the seven-block instance, dgrad 160 -> 96;
    the level-0 downsample branch 128 -> 96 stays on k_conv_gather (as fast there: tools/dbg/pointwise_ab.py)"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([4], voxel=0.02, n_target=80000)
    assert coords.shape[0] >= 66000
    feats = np.random.default_rng(5).standard_normal((coords.shape[0], cin)).astype(np.float32)
    if dtype == torch.bfloat16:
        feats = torch.from_numpy(feats).bfloat16().float().numpy()
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=1, stride=1, bias=bias, dimension=3)], coords, feats, dtype=dtype,
        oracle_impl="torch")
    assert rel_err(h_out, o_out) < tol
    for n, a, b in zip(["dgrad", "wgrad", "bgrad"], h_g, o_g):
        assert rel_err(a, b) < tol * (5 if dtype == torch.float32 else 1), n


# ------------------------------------------------------------------------------------------- BatchNorm eval cache
def _bn_eval_ref(bn, x):
    with torch.no_grad():
        return torch.nn.functional.batch_norm(x.detach().float(), bn.bn.running_mean, bn.bn.running_var, bn.bn.weight, bn.bn.bias, False, 0.0, bn.bn.eps)


@pytest.mark.parity("plain torch restatement")
@pytest.mark.parametrize("sync", [False, True])
def test_eval_statistics_follow_training_updates(sync):
    """train -> eval -> train -> eval on one MinkowskiBatchNorm: a Lightning-style sanity validation BEFORE training caches
    [mean | invstd]; the engine then updates running_mean / running_var through raw pointers (lgs_bn_forward), which torch's
    version counters did not see, so the second validation silently reused the first one's statistics."""
    torch.manual_seed(2)
    n, c = 6001, 64
    coords = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.arange(n, dtype=torch.int32)[:, None].repeat(1, 3)], 1).to(DEV)
    xs = ME.SparseTensor(torch.randn(n, c, device=DEV) * 3 + 1.5, coords)
    bn = (ME.MinkowskiSyncBatchNorm if sync else ME.MinkowskiBatchNorm)(c, momentum=0.5).to(DEV)
    bn.eval()
    y0 = bn(xs).F.detach()                                     # sanity validation: running stats are still (0, 1)
    assert rel_err(y0.cpu().numpy(), _bn_eval_ref(bn, xs.F).cpu().numpy()) < 1e-5
    v0 = bn.bn.running_mean._version
    bn.train()
    for _ in range(2):
        bn(xs)                                                 # the engine moves the running statistics
    assert bn.bn.running_mean._version > v0                    # ... and says so
    assert float(bn.bn.running_mean.abs().max()) > 0.5
    bn.eval()
    y1 = bn(xs).F.detach()
    assert rel_err(y1.cpu().numpy(), _bn_eval_ref(bn, xs.F).cpu().numpy()) < 1e-5
    assert not torch.allclose(y0, y1)
    bn.train(); bn(xs); bn.eval()
    y2 = bn(xs).F.detach()
    assert rel_err(y2.cpu().numpy(), _bn_eval_ref(bn, xs.F).cpu().numpy()) < 1e-5


def test_model_validation_after_training_uses_fresh_statistics():
    """the same through the whole-block fast path (it calls lgs_bn_forward itself): eval logits of Res16UNet14A after two
    training steps equal those of a copy whose eval caches never existed"""
    import copy
    from languagegroundedsemseg_amd import models
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, feats, labels = make_batch([0], voxel=0.05, n_target=20000)
    m = deterministic_init(models.load_model("Res16UNet14A")(3, 20, Cfg()), 42).to(DEV)
    x = lambda: ME.SparseTensor(torch.from_numpy(feats).to(DEV), torch.from_numpy(coords).to(DEV))
    m.eval()
    with torch.no_grad():
        before = m(x())[0].F.clone()
    m.train()
    for _ in range(2):
        m(x())[0].F.float().sum().backward()
    m.eval()
    with torch.no_grad():
        after = m(x())[0].F.clone()
    fresh = models.load_model("Res16UNet14A")(3, 20, Cfg()).to(DEV)
    fresh.load_state_dict(copy.deepcopy(m.state_dict()))
    fresh.eval()
    with torch.no_grad():
        want = fresh(x())[0].F
    assert torch.equal(after, want)
    assert not torch.allclose(after, before)


# ------------------------------------------------------------------------------------------- block fast path guard
def test_block_fast_path_declines_other_conv_shapes():
    from languagegroundedsemseg_amd.me import deferred
    from languagegroundedsemseg_amd.models import BasicBlock
    coords = torch.cat([torch.zeros(500, 1, dtype=torch.int32), torch.randint(0, 12, (500, 3), dtype=torch.int32)], 1).unique(dim=0).to(DEV)
    x = ME.SparseTensor(torch.randn(coords.shape[0], 32, device=DEV), coords)
    blk = BasicBlock(32, 32, D=3).to(DEV).train()
    n0 = deferred.STATS["blocks"]
    assert torch.isfinite(blk(x).F).all() and deferred.STATS["blocks"] == n0 + 1     # the recorded calls ran as ONE node
    odd = BasicBlock(32, 32, D=3)
    odd.conv2 = ME.MinkowskiConvolution(32, 32, kernel_size=1, stride=1, dimension=3)
    odd = odd.to(DEV).train()
    y = odd(x)                                                  # the call-by-call path computes it (1x1 second conv)
    assert y.F.shape == x.F.shape and torch.isfinite(y.F).all() and deferred.STATS["blocks"] == n0 + 1


# ------------------------------------------------------------------------------------------- C-side block entry points
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("with_ddp", [True, False])
@pytest.mark.parametrize("inline", [True, False])
def test_c_side_block_equals_the_call_by_call_block(monkeypatch, dtype, with_ddp, inline):
    """lgs_block_forward / lgs_block_backward (one engine call per BasicBlock and direction) issue the launches of the call-by-call
    path in the same order -- weight gradients on the compute stream (inline: the small-batch regime) or, from inside the engine
    call, on the side stream behind a fork event with the parameter's event recorded after them (what BucketedDDP waits for) --:
    logits, every gradient (through the gradient-bucket slots and without them) and the running statistics are bit-identical; Res16UNet34C has blocks with and without the 1x1 downsample branch and
    without a final ReLU is covered by 34D's last block.  Reference: /root/reference/models/modules/resnet_block.py:41-57."""
    from languagegroundedsemseg_amd import engine, models
    from languagegroundedsemseg_amd.ddp import BucketedDDP
    from languagegroundedsemseg_amd.losses import fused_cross_entropy
    from languagegroundedsemseg_amd.me import backend_hip, block as _block
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, feats, labels = make_batch([8], voxel=0.05, n_target=9000)
    c, f = torch.from_numpy(coords).to(DEV), torch.from_numpy(feats).to(DEV).to(dtype)
    l = torch.from_numpy(labels % 20).to(DEV)
    monkeypatch.setattr(backend_hip, "_WGRAD_INLINE_BELOW", (1 << 30) if inline else 0)

    def run(c_path, name):
        monkeypatch.setattr(_block, "_BLOCK_C", c_path)
        be = ME.get_backend()
        calls0 = getattr(be, "block_calls", 0)
        m = deterministic_init(models.load_model(name)(3, 20, Cfg()), 11).to(DEV).train()
        if name == "Res16UNet34D":
            m.representation_only(True)
        ddp = BucketedDDP(m, bucket_mb=1.0) if with_ddp else None
        outs = []
        for step in range(2):                                   # second step: running statistics of the first feed nothing, but must match
            if ddp is not None:
                ddp.zero_grad()
            else:
                m.zero_grad(set_to_none=True)
            x = ME.SparseTensor(f, c)
            assert bool(x.coordinate_manager._m.inline_wgrad) == inline
            engine.dispatch_counts(reset=True)
            y = m(x)
            out = y[0].F if isinstance(y, tuple) else y.F
            loss = fused_cross_entropy(out[:, :20].contiguous(), l, ignore_index=-1) if out.shape[1] >= 20 else out.float().square().mean()
            loss.backward()
            if ddp is not None:
                ddp.finalize()
            torch.cuda.synchronize()
            outs.append((out.detach().float().cpu().clone(),
                         {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None},
                         {k: b.detach().float().cpu().clone() for k, b in m.named_buffers()}))
        n_blocks = sum(1 for mod in m.modules() if isinstance(mod, models.BasicBlock))
        assert getattr(be, "block_calls", 0) - calls0 == (2 * 2 * n_blocks if c_path else 0)     # fwd + bwd, two steps
        return outs

    for name in ("Res16UNet34C", "Res16UNet34D") if dtype == torch.bfloat16 else ("Res16UNet14A",):
        a, b = run(True, name), run(False, name)
        for (oa, ga, ba), (ob, gb, bb) in zip(a, b):
            assert torch.equal(oa, ob), name
            assert set(ga) == set(gb) and len(ga) > 50
            for k in ga:
                assert torch.equal(ga[k], gb[k]), (name, k)
            for k in ba:
                assert torch.equal(ba[k], bb[k]), (name, k)


@pytest.mark.gpu
@pytest.mark.parity("plain torch restatement")
@pytest.mark.parametrize("n,c,ignore", [(100003, 200, -1), (4097, 20, 255), (1, 13, -1), (70001, 200, 7)])
def test_cross_entropy_denominator_is_counted_on_the_device(n, c, ignore):
    """lgs_ce_count_valid = the rows nn.CrossEntropyLoss(ignore_index) 'mean' divides by (pl_BaselineTrainer.py:350): labels equal to
    ignore_index or outside [0, C) do not count -- including an ignore_index that collides with a real class -- and the fused loss
    (one count launch, no elementwise chain over the labels) equals torch's on the same rows, forward and backward."""
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.losses import fused_cross_entropy
    from languagegroundedsemseg_amd.me.backend_hip import _ptr, _stream
    g = torch.Generator().manual_seed(n)
    labels = torch.randint(0, c, (n,), generator=g)
    labels[torch.rand(n, generator=g) < 0.15] = ignore
    labels[torch.rand(n, generator=g) < 0.02] = c + 5          # outside the head: ignored rows, not errors
    labels[torch.rand(n, generator=g) < 0.02] = -7
    want = int(((labels != ignore) & (labels >= 0) & (labels < c)).sum())
    lab = labels.cuda()
    cnt = torch.full((1,), 123, dtype=torch.int32, device="cuda")
    with torch.cuda.device(0):
        engine.check(engine.lib().lgs_ce_count_valid(_ptr(lab), n, c, ignore, _ptr(cnt), _stream()))
    assert int(cnt.item()) == want
    logits = torch.randn(n, c, generator=g).cuda().requires_grad_(True)
    loss = fused_cross_entropy(logits, lab, ignore_index=ignore)
    (loss * 3.0).backward()
    ref_in = logits.detach().clone().requires_grad_(True)
    safe = torch.where((lab >= 0) & (lab < c) & (lab != ignore), lab, torch.full_like(lab, -100))
    ref = torch.nn.functional.cross_entropy(ref_in, safe, ignore_index=-100) if want else ref_in.sum() * 0
    (ref * 3.0).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert float((logits.grad - ref_in.grad).abs().max()) <= 1e-6
