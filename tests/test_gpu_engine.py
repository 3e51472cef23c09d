"""GPU parity tests proper: the HIP engine (through the C-ABI, via the MinkowskiEngine-compatible
surface) against the CPU oracle on identical voxelised inputs.
Bars (north_star): coordinate / kernel maps bit-exact as sets; fp32 features within 1e-3 (we hold 2e-4);
bf16 storage path reported against the fp32 oracle with a bf16-sized tolerance."""
import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from helpers import canon, small_scene, triples_as_set
from oracle import oracle as orc
from oracle.backend import OracleBackend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_tensor(feats, coords, dtype=torch.float32):
    return ME.SparseTensor(torch.from_numpy(feats).to(DEV).to(dtype), torch.from_numpy(coords).to(DEV))


def test_native_library_loaded():
    from languagegroundedsemseg_amd import engine
    assert engine.lib().lgs_abi_version() == engine.ABI_VERSION
    assert ME.get_backend().name == "hip"


# ------------------------------------------------------------------------------------------- maps
def test_insert_dedup_matches_oracle_bit_exact():
    rng = np.random.default_rng(0)
    c = small_scene(1, n=3000)
    c = np.concatenate([c, c[rng.integers(0, c.shape[0], 700)]], 0)     # duplicates
    c = c[rng.permutation(c.shape[0])]
    c[:, 1:] += np.array([-300, 57, 1000], np.int32)                    # negative + shifted
    f = rng.standard_normal((c.shape[0], 3)).astype(np.float32)
    x = hip_tensor(f, c)
    ui, inv = orc.unique_coords(c)
    assert np.array_equal(x.unique_index.cpu().numpy(), ui)
    assert np.array_equal(x.inverse_mapping.cpu().numpy(), inv)
    assert np.array_equal(x.C.cpu().numpy(), c[ui])
    assert np.array_equal(x.F.cpu().numpy(), f[ui])


@pytest.mark.parametrize("seed", [2, 3])
def test_strided_maps_and_kernel_maps_equal_as_sets(seed):
    c = small_scene(seed, n=4000, extent=40)
    c[:, 1:] -= 17
    x = hip_tensor(np.zeros((c.shape[0], 3), np.float32), c)
    mgr = x.coordinate_manager
    keys, ocoords, ts = [x.coordinate_map_key], [c], 1
    for lvl in range(4):
        keys.append(mgr.stride(keys[-1], 2))
        oc, _ = orc.stride_coords(ocoords[-1], ts * 2)
        ts *= 2
        ocoords.append(oc)
        hc = mgr.get_coordinates(keys[-1]).cpu().numpy()
        assert hc.shape == oc.shape
        assert np.array_equal(hc[canon(hc)], oc[canon(oc)])
    # the coarse maps were sized from the counts the insert took in its own pass (no synchronisation): every builder's own run
    # count agreed with them (lgs_manager_check: device-side flag, advisor round 4)
    assert mgr._m.check() == 0
    ts = 1
    for lvl in range(5):
        hc = mgr.get_coordinates(keys[lvl]).cpu().numpy()
        k, i, o = [t.cpu().numpy() for t in mgr.kernel_map_handle(keys[lvl], keys[lvl], 3).export()]
        ok, oi, oo = orc.kernel_map(ocoords[lvl], ocoords[lvl], 3, ts)
        assert k.shape[0] == ok.shape[0]
        assert np.array_equal(triples_as_set(hc, hc, k, i, o), triples_as_set(ocoords[lvl], ocoords[lvl], ok, oi, oo))
        if lvl < 4:
            hco = mgr.get_coordinates(keys[lvl + 1]).cpu().numpy()
            k, i, o = [t.cpu().numpy() for t in mgr.kernel_map_handle(keys[lvl], keys[lvl + 1], 2).export()]
            ok, oi, oo = orc.kernel_map(ocoords[lvl], ocoords[lvl + 1], 2, ts)
            assert k.shape[0] == hc.shape[0] == ok.shape[0]            # one pair per fine voxel
            assert np.array_equal(triples_as_set(hc, hco, k, i, o),
                                  triples_as_set(ocoords[lvl], ocoords[lvl + 1], ok, oi, oo))
        ts *= 2


def test_out_of_range_coordinates_fail_loudly():
    c = np.array([[0, 0, 0, 0], [0, 200000, 0, 0]], np.int32)
    with pytest.raises(RuntimeError, match="out of range"):
        hip_tensor(np.zeros((2, 3), np.float32), c)


# ------------------------------------------------------------------------------------------- conv
def run_both(build, coords, feats, dtype=torch.float32, grad_seed=0, oracle_impl="c"):
    """run the same module graph on the HIP engine and on the oracle; return canonicalised outputs + grads"""
    res = []
    for backend in ("hip", "oracle"):
        prev = ME.get_backend()
        if backend == "oracle":
            ME.set_backend(OracleBackend(oracle_impl))
        try:
            torch.manual_seed(1)
            mods = build()
            dev = DEV if backend == "hip" else "cpu"
            mods = [m.to(dev) for m in mods]
            f = torch.from_numpy(feats).to(dev)
            f = f.to(dtype) if backend == "hip" else f
            f.requires_grad_(True)
            x = ME.SparseTensor(f, torch.from_numpy(coords).to(dev))
            y = x
            for m in mods:
                y = m(y)
            perm = canon(y.C.cpu().numpy())
            g = torch.from_numpy(np.random.default_rng(grad_seed).standard_normal(tuple(y.F.shape)).astype(np.float32))
            inv = np.empty_like(perm)
            inv[perm] = np.arange(perm.shape[0])
            gy = g[torch.from_numpy(inv)].to(dev).to(y.F.dtype)        # same gradient per COORDINATE on both sides
            y.F.backward(gy)
            out = y.F.detach().float().cpu().numpy()[perm]
            grads = [f.grad.float().cpu().numpy()] + [p.grad.float().cpu().numpy() for m in mods for p in m.parameters()]
            res.append((out, grads))
        finally:
            ME.set_backend(prev)
    return res


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


CONV_CASES = [
    # (cin, cout, kernel, stride, transposed_after_down, bias)
    (3, 32, 3, 1, False, False),
    (32, 32, 3, 1, False, False),
    (64, 96, 3, 1, False, False),
    (128, 96, 3, 1, False, False),
    (96, 200, 1, 1, False, True),
    (32, 64, 2, 2, False, False),
    (40, 24, 3, 1, False, False),      # channel counts that are not multiples of 32
]


@pytest.mark.parametrize("cin,cout,ks,st,tr,bias", CONV_CASES)
def test_conv_fwd_bwd_fp32_matches_oracle(cin, cout, ks, st, tr, bias):
    coords = small_scene(11, n=2500, extent=30)
    feats = np.random.default_rng(5).standard_normal((coords.shape[0], cin)).astype(np.float32)
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=st, bias=bias, dimension=3)], coords, feats)
    assert h_out.shape == o_out.shape
    assert rel_err(h_out, o_out) < 2e-5
    names = ["dgrad", "wgrad", "bgrad"]
    for n, a, b in zip(names, h_g, o_g):
        if n == "dgrad" and cin % 4 != 0:
            continue
        assert rel_err(a, b) < 5e-5, n


def test_conv0_input_grad_is_not_required():
    """conv0p1s1 (3 -> 32) never needs dgrad: the network input carries no gradient."""
    coords = small_scene(12, n=800)
    x = hip_tensor(np.ones((coords.shape[0], 3), np.float32), coords)
    conv = ME.MinkowskiConvolution(3, 32, kernel_size=3, dimension=3).to(DEV)
    y = conv(x)
    y.F.sum().backward()
    assert conv.kernel.grad is not None and torch.isfinite(conv.kernel.grad).all()


def test_down_then_transposed_up_matches_oracle():
    coords = small_scene(13, n=3000, extent=32)
    feats = np.random.default_rng(6).standard_normal((coords.shape[0], 32)).astype(np.float32)

    def build():
        return [ME.MinkowskiConvolution(32, 64, kernel_size=2, stride=2, dimension=3),
                ME.MinkowskiConvolution(64, 64, kernel_size=3, stride=1, dimension=3),
                ME.MinkowskiConvolutionTranspose(64, 96, kernel_size=2, stride=2, dimension=3)]
    (h_out, h_g), (o_out, o_g) = run_both(build, coords, feats)
    assert rel_err(h_out, o_out) < 5e-5
    for a, b in zip(h_g, o_g):
        assert rel_err(a, b) < 1e-4


def test_conv_bf16_storage_against_fp32_oracle():
    coords = small_scene(14, n=2500, extent=30)
    feats = np.random.default_rng(7).standard_normal((coords.shape[0], 64)).astype(np.float32)
    feats = torch.from_numpy(feats).bfloat16().float().numpy()          # inputs exactly representable
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(64, 96, kernel_size=3, stride=1, dimension=3)], coords, feats, dtype=torch.bfloat16)
    assert rel_err(h_out, o_out) < 2e-2       # bf16 weights + bf16 output rounding
    assert rel_err(h_g[0], o_g[0]) < 2e-2
    assert rel_err(h_g[1], o_g[1]) < 2e-2


def test_single_voxel_and_ragged_batches():
    coords = np.array([[0, 5, 5, 5]], np.int32)
    x = hip_tensor(np.ones((1, 32), np.float32), coords)
    conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, dimension=3).to(DEV)
    y = conv(x)
    assert torch.allclose(y.F, conv.kernel[13].sum(0, keepdim=True).to(y.F.dtype), atol=1e-5)
    # batch index 1 is empty, 0 and 2 populated
    c = small_scene(15, n=600, batches=2)
    c[c[:, 0] == 1, 0] = 2
    f = np.random.default_rng(1).standard_normal((c.shape[0], 32)).astype(np.float32)
    (h_out, _), (o_out, _) = run_both(lambda: [ME.MinkowskiConvolution(32, 32, kernel_size=3, dimension=3)], c, f)
    assert rel_err(h_out, o_out) < 2e-5


# ------------------------------------------------------------------------------------------- norm
@pytest.mark.parity("plain torch: nn.BatchNorm1d / F.cross_entropy + autograd")
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("launches", ["default", "one", "three"])
def test_fused_bn_matches_torch(dtype, tol, relu, res, launches):
    """launches: the default policy (forward = column sums / fold / apply, backward = one grid-barrier launch for layers <= 24 MB),
    the one-launch kernels in BOTH directions, three launches in both"""
    from languagegroundedsemseg_amd import engine
    knobs = {"default": {}, "one": dict(BN_FUSED=1, BN_FUSED_FWD_MAX_MB=24), "three": dict(BN_FUSED=0)}[launches]
    with engine.tuning(**knobs):
        _bn_matches_torch(dtype, tol, relu, res)


def _bn_matches_torch(dtype, tol, relu, res):
    torch.manual_seed(0)
    n, c = 5000, 96
    xf = torch.randn(n, c) * 2 + 0.5
    rf = torch.randn(n, c)
    if dtype == torch.bfloat16:
        xf, rf = xf.bfloat16().float(), rf.bfloat16().float()
    coords = small_scene(16, n=20000, extent=64)[:n]
    assert coords.shape[0] == n
    bn_h = ME.MinkowskiBatchNorm(c, momentum=0.02).to(DEV)
    bn_t = torch.nn.BatchNorm1d(c, momentum=0.02)
    with torch.no_grad():
        bn_h.bn.weight.copy_(torch.rand(c) + 0.5); bn_h.bn.bias.copy_(torch.randn(c) * 0.1)
        bn_t.weight.copy_(bn_h.bn.weight.cpu()); bn_t.bias.copy_(bn_h.bn.bias.cpu())
    xh = xf.to(DEV).to(dtype).requires_grad_(True)
    rh = rf.to(DEV).to(dtype).requires_grad_(True)
    sx = ME.SparseTensor(xh, torch.from_numpy(coords).to(DEV))
    sr = ME.SparseTensor(rh, coordinate_map_key=sx.coordinate_map_key, coordinate_manager=sx.coordinate_manager)
    y = bn_h(sx, relu=relu, residual=sr if res else None).F
    xt = xf.clone().requires_grad_(True)
    rt = rf.clone().requires_grad_(True)
    yt = bn_t(xt)
    if res:
        yt = yt + rt
    if relu:
        yt = torch.relu(yt)
    g = torch.randn(n, c)
    y.backward(g.to(DEV).to(dtype))
    yt.backward(g)
    assert rel_err(y.detach().float().cpu().numpy(), yt.detach().numpy()) < tol
    assert rel_err(xh.grad.float().cpu().numpy(), xt.grad.numpy()) < tol * 5
    if res:
        assert rel_err(rh.grad.float().cpu().numpy(), rt.grad.numpy()) < tol
    assert rel_err(bn_h.bn.weight.grad.cpu().numpy(), bn_t.weight.grad.numpy()) < tol * 5
    assert rel_err(bn_h.bn.bias.grad.cpu().numpy(), bn_t.bias.grad.numpy()) < tol * 5
    assert rel_err(bn_h.bn.running_mean.cpu().numpy(), bn_t.running_mean.numpy()) < max(tol, 1e-4)
    assert rel_err(bn_h.bn.running_var.cpu().numpy(), bn_t.running_var.numpy()) < max(tol, 1e-4)
    assert int(bn_h.bn.num_batches_tracked) == int(bn_t.num_batches_tracked) == 1   # counted inside the fold kernel


# ------------------------------------------------------------------------------------------- CLIP
@pytest.mark.parity("fp64 restatement of normalize(F) . normalize(T)^T")
@pytest.mark.parametrize("c", [512, 96])
def test_clip_similarity_mfma_matches_fp64(c):
    torch.manual_seed(3)
    f = torch.randn(3000, c)
    t = torch.randn(200, c)
    sim, inv = ME.get_backend().clip_similarity(f.to(DEV), t.to(DEV))
    ref, rinv = OracleBackend().clip_similarity(f, t)
    assert rel_err(sim.cpu().numpy(), ref.numpy()) < 2e-5
    assert rel_err(inv.cpu().numpy(), rinv.numpy()) < 1e-5
    simb, _ = ME.get_backend().clip_similarity(f.to(DEV).bfloat16(), t.to(DEV))
    assert np.abs(simb.cpu().numpy() - ref.numpy()).max() < 6e-3


BF16_CASES = [(32, 32, 3, 1), (96, 96, 3, 1), (128, 96, 3, 1), (256, 128, 3, 1), (192, 128, 3, 1), (96, 200, 1, 1),
              (64, 64, 2, 2), (384, 256, 3, 1), (8, 16, 3, 1)]


@pytest.mark.parametrize("cin,cout,ks,st", BF16_CASES)
def test_bf16_mfma_paths_all_tile_shapes(cin, cout, ks, st):
    """every (wave tile, channel split) instance of the bf16 forward / dgrad / wgrad kernels vs the fp32 oracle"""
    coords = small_scene(21, n=3000, extent=28)
    feats = np.random.default_rng(9).standard_normal((coords.shape[0], cin)).astype(np.float32)
    feats = torch.from_numpy(feats).bfloat16().float().numpy()
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=st, bias=(ks == 1), dimension=3)], coords, feats,
        dtype=torch.bfloat16)
    assert rel_err(h_out, o_out) < 2e-2
    for n, a, b in zip(["dgrad", "wgrad", "bgrad"], h_g, o_g):
        assert rel_err(a, b) < 2e-2, n


@pytest.mark.parametrize("cin,cout", [(96, 96), (128, 96), (32, 32), (32, 64), (64, 256), (256, 512), (512, 256)])
def test_bf16_conv_on_a_large_map(cin, cout):
    """maps of >= 65536 positions take the 'big' tile configurations of k_conv_gather and the one-offset-per-wave
    schedule of k_wgrad_bf16; reading y.C while kernels are still queued must not disturb them (the coordinate export
    once wrote into a just-released conv workspace from the map stream).  256 -> 512 and 512 -> 256 are the wide-channel
    shapes of Res16UNet34D: eight-wave interleaved conv tiles with two column tiles, and the weight gradient's three-block
    stationary slices with a partly padded last slice (16 blocks -> 6 slices; 8 blocks x 16 gathered slices -> 3 slices)"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([3], voxel=0.02, n_target=80000)
    assert coords.shape[0] >= 66000
    feats = torch.from_numpy(np.random.default_rng(5).standard_normal((coords.shape[0], cin)).astype(np.float32)).bfloat16().float().numpy()
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3)], coords, feats, dtype=torch.bfloat16,
        oracle_impl="torch")
    assert rel_err(h_out, o_out) < 2e-2
    for n, a, b in zip(["dgrad", "wgrad"], h_g, o_g):
        assert rel_err(a, b) < 2e-2, n


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (32, 32), (96, 96), (64, 128)])
def test_fp32_conv_on_a_large_map(cin, cout):
    """the fp32 (parity-path) tiles of maps >= 65536 positions: 1 / 2 / 3 / 4 column blocks per workgroup -- the 64-channel layers of
    level 2 in the benchmark's 8-scene batch take the two-block tile, which no small-scene test reaches (found by the tightened
    dispatch-coverage assertion, round 5); forward, dgrad and weight gradient against the oracle"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([3], voxel=0.02, n_target=80000)
    assert coords.shape[0] >= 66000
    feats = np.random.default_rng(5).standard_normal((coords.shape[0], cin)).astype(np.float32)
    (h_out, h_g), (o_out, o_g) = run_both(
        lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3)], coords, feats, oracle_impl="torch")
    assert rel_err(h_out, o_out) < 2e-5
    for n, a, b in zip(["dgrad", "wgrad"], h_g, o_g):
        assert rel_err(a, b) < 1e-4, n


def test_bf16_transposed_conv_wgrad():
    coords = small_scene(22, n=3000, extent=32)
    feats = torch.from_numpy(np.random.default_rng(3).standard_normal((coords.shape[0], 32)).astype(np.float32)).bfloat16().float().numpy()

    def build():
        return [ME.MinkowskiConvolution(32, 64, kernel_size=2, stride=2, dimension=3),
                ME.MinkowskiConvolutionTranspose(64, 96, kernel_size=2, stride=2, dimension=3)]
    (h_out, h_g), (o_out, o_g) = run_both(build, coords, feats, dtype=torch.bfloat16)
    assert rel_err(h_out, o_out) < 3e-2
    for a, b in zip(h_g, o_g):
        assert rel_err(a, b) < 3e-2


# ------------------------------------------------------------------------------------------- losses
@pytest.mark.parity("plain torch: nn.BatchNorm1d / F.cross_entropy + autograd")
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_fused_cross_entropy_matches_torch(dtype, tol):
    from languagegroundedsemseg_amd.losses import fused_cross_entropy
    torch.manual_seed(0)
    x = (torch.randn(7001, 200) * 3).to(dtype).float()
    lab = torch.randint(-1, 200, (7001,))
    xh = x.to(DEV).to(dtype).requires_grad_(True)
    loss = fused_cross_entropy(xh, lab.to(DEV), -1)
    (loss * 2.0).backward()
    xt = x.clone().requires_grad_(True)
    lt = torch.nn.functional.cross_entropy(xt, lab, ignore_index=-1)
    (lt * 2.0).backward()
    assert abs(float(loss) - float(lt)) < 1e-4
    assert rel_err(xh.grad.float().cpu().numpy(), xt.grad.numpy()) < tol


def test_contrastive_loss_on_mfma_matches_reference_golden():
    import os
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "contrastive_loss.npz"))
    for tag in ("c512", "c96"):
        g = lambda k: torch.from_numpy(fx["%s_%s" % (tag, k)]).to(DEV)
        crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
        F = g("F").clone().requires_grad_(True)
        loss, pos, neg = crit(F, g("labels"), g("T"), neg_indices=g("neg"))
        assert torch.allclose(pos, g("pos_loss"), atol=1e-5)
        assert torch.allclose(neg, g("neg_loss"), atol=1e-5)
        assert abs(float(loss) - float(g("total"))) < 1e-5
        loss.backward()
        assert torch.isfinite(F.grad).all()


@pytest.mark.parity("plain torch: nn.BatchNorm1d / F.cross_entropy + autograd")
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_backward_reads_a_column_slice_in_place(dtype):
    """the gradient of one ME.cat input is a column slice of a wider tensor: lgs_bn_backward(dy_row_stride) must give
    bit-identical results to the contiguous copy"""
    be = ME.get_backend()
    torch.manual_seed(3)
    n, c, wide = 4097, 96, 128
    x = torch.randn(n, c, device=DEV).to(dtype)
    g1, b1 = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.1
    y, st = be.bn_forward(x, g1, b1, 1e-5, 0.1, None, None, None, 1)
    big = torch.randn(n, wide, device=DEV).to(dtype)
    for off in (0, 32):
        dy = big[:, off:off + c]
        assert not dy.is_contiguous()
        a = be.bn_backward(x, None, dy, g1, b1, st, 2, False)
        b = be.bn_backward(x, None, dy.contiguous(), g1, b1, st, 2, False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
