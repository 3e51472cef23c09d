"""k_conv_halo (csrc/lgs_conv_halo.hip): the per-tile distinct-row 3^3 convolution that bf16 layers of <= 128 channels take on
maps that carry halo tables -- with the tuning knob HALO = 1 (default 0: measured slower than k_conv_gather at the benchmark's
shapes; the kernel stays parity-tested so that the measurement can be repeated).  Held (a) to the fp32 oracle on bf16-rounded inputs like every bf16 per-op test (2e-2 max-norm),
(b) to the k_conv_gather path on the same inputs (same bf16 products, another summation order: <= 2e-3 rel-L2, i.e. bf16
rounding flips only), on every instantiated (gathered-chunk, output-block) shape, forward and dgrad, on
  * a 2 cm surface scene of >= 65 k voxels (production gate HALO_MIN_ROWS),
  * DENSE random volumes, whose tiles have far more distinct rows than the LDS row buffer holds: list segments (count > CAP)
    and per-offset staging (count = -1),
  * tiny / ragged maps (gate lowered), a strided (zero-copy cat) input and the accumulating dgrad epilogue.
Reference: /root/reference/models/modules/common.py:179-203, models/modules/resnet_block.py:41-57."""
import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from helpers import small_scene
from test_gpu_engine import rel_err, run_both

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(96, 96), (128, 96), (96, 128), (32, 32), (64, 64), (32, 64), (64, 32), (128, 128), (192, 128), (64, 96), (64, 128), (40, 24), (96, 160)]


def _halo_shape_ok(g, o):
    """mirror of halo_shape() in csrc/lgs_conv_halo.hip: gathered channels g -> (chunks per pass NC), outputs o -> blocks per pass"""
    if g % 8 or o % 4 or g < 32 or o < 8 or o > 128:
        return False
    gc, nb = (g + 31) // 32, (o + 31) // 32
    nc = gc if gc <= 3 else (2 if gc == 4 else (3 if gc == 6 else 0))
    nbp = 2 if nb == 4 else nb
    return (nc == 1 and nbp in (1, 2)) or (nc == 2 and 1 <= nbp <= 3) or (nc == 3 and nbp in (2, 3))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def halo_hits(engine):
    return {k: v for k, v in engine.dispatch_counts().items() if k.startswith("k_conv_halo")}


def _conv_both_paths(coords, cin, cout, seed=0, min_rows=None):
    """forward + dgrad of one 3^3 conv through the autograd surface: (halo out, halo dx, gather out, gather dx, halo launches)"""
    from languagegroundedsemseg_amd import engine
    torch.manual_seed(seed)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3).to(DEV)
    c = torch.from_numpy(coords).to(DEV)
    f0 = torch.randn(coords.shape[0], cin, device=DEV).bfloat16()
    g0 = torch.randn(coords.shape[0], cout, device=DEV).bfloat16()
    res = []
    for halo in (1, 0):
        knobs = {"HALO": halo}
        if min_rows is not None:
            knobs["HALO_MIN_ROWS"] = min_rows
        with engine.tuning(**knobs):
            f = f0.clone().requires_grad_(True)
            x = ME.SparseTensor(f, c)
            engine.dispatch_counts(reset=True)
            y = conv(x)
            assert torch.equal(y.C, x.C)
            y.F.backward(g0)
            torch.cuda.synchronize()
            res.append((y.F.detach().float().cpu().numpy(), f.grad.float().cpu().numpy(), halo_hits(engine)))
        ME.get_backend().invalidate_packed_weights()
    return res


@pytest.mark.parametrize("cin,cout", SHAPES)
def test_halo_conv_matches_gather_path_and_oracle_on_a_surface_scene(cin, cout):
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([3], voxel=0.02, n_target=80000)
    assert coords.shape[0] >= 66000
    (ho, hd, hits), (go, gd, nohits) = _conv_both_paths(coords, cin, cout)
    # forward + dgrad on the halo kernel; outputs wider than 128 channels stay on k_conv_gather (standard view): 96 -> 160 not at
    # all, 192 -> 128 in the forward direction only (its dgrad writes 192 channels)
    from languagegroundedsemseg_amd.me.backend_hip import _ptr  # noqa: F401
    want = (1 if _halo_shape_ok(cin, cout) else 0) + (1 if _halo_shape_ok(cout, cin) else 0)
    assert sum(hits.values()) == want, hits
    assert not nohits
    assert rel_l2(ho, go) < 2e-3 and rel_l2(hd, gd) < 2e-3, (rel_l2(ho, go), rel_l2(hd, gd))
    if (cin, cout) in ((96, 96), (128, 96), (32, 32)):           # and against the oracle (the other shapes: via the gather path above)
        feats = torch.from_numpy(np.random.default_rng(5).standard_normal((coords.shape[0], cin)).astype(np.float32)).bfloat16().float().numpy()
        engine.dispatch_counts(reset=True)
        with engine.tuning(HALO=1):
            (h_out, h_g), (o_out, o_g) = run_both(
                lambda: [ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, dimension=3)], coords, feats, dtype=torch.bfloat16,
                oracle_impl="torch")
        assert sum(halo_hits(engine).values()) == 2
        assert rel_err(h_out, o_out) < 2e-2
        for n, a, b in zip(["dgrad", "wgrad"], h_g, o_g):
            assert rel_err(a, b) < 2e-2, n


def dense_volume(n_side, density, seed, batches=2):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batches):
        m = rng.random((n_side, n_side, n_side)) < density
        p = np.argwhere(m).astype(np.int32) - n_side // 2
        p = p[rng.permutation(p.shape[0])]
        out.append(np.concatenate([np.full((p.shape[0], 1), b, np.int32), p], 1))
    return np.concatenate(out, 0)


@pytest.mark.parametrize("density,cin,cout", [(0.9, 96, 96), (0.5, 96, 96), (0.5, 128, 96), (0.15, 32, 32), (0.9, 64, 64), (0.02, 96, 96)])
def test_halo_conv_on_dense_and_sparse_random_volumes(density, cin, cout):
    """a dense 3-D block has ~600-2000 distinct rows per 256-position tile (a surface: ~380): the row list no longer fits the
    LDS row buffer (512 rows at 96 channels) -> list segments, and above 1024 -> per-offset staging; a 2 % volume is the
    opposite extreme (almost no neighbours, most (block, offset) pairs skipped)"""
    coords = dense_volume(44 if density >= 0.5 else (56 if density >= 0.1 else 120), density, seed=int(density * 100))
    assert coords.shape[0] >= 30000
    (ho, hd, hits), (go, gd, _) = _conv_both_paths(coords, cin, cout, min_rows=0)
    assert sum(hits.values()) == 2, hits
    assert rel_l2(ho, go) < 2e-3 and rel_l2(hd, gd) < 2e-3, (rel_l2(ho, go), rel_l2(hd, gd))


@pytest.mark.parametrize("n", [1, 37, 255, 256, 257, 1500])
def test_halo_conv_on_tiny_and_ragged_maps(n):
    coords = small_scene(n, n=6 * n + 16, extent=24, batches=1)[:n]
    assert coords.shape[0] == n
    (ho, hd, hits), (go, gd, _) = _conv_both_paths(coords, 32, 32, min_rows=0)
    assert sum(hits.values()) == 2, hits
    assert ho.shape == go.shape == (n, 32)
    assert rel_l2(ho, go) < 2e-3 and rel_l2(hd, gd) < 2e-3


def test_halo_conv_reads_a_strided_input_and_accumulates_in_the_epilogue():
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([5], voxel=0.02, n_target=80000)
    n = coords.shape[0]
    engine.tuning_set("HALO", 1)
    x = ME.SparseTensor(torch.zeros(n, 3, device=DEV).bfloat16(), torch.from_numpy(coords).to(DEV))
    km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
    g = torch.Generator(device=DEV).manual_seed(1)
    big = torch.randn(n, 128, device=DEV, generator=g).bfloat16()
    w = torch.randn(27, 96, 96, device=DEV, generator=g) * 0.05
    sl = big[:, 32:128]
    engine.dispatch_counts(reset=True)
    a = km.conv_forward(sl, w, None, False)
    b = km.conv_forward(sl.contiguous(), w, None, False)
    assert sum(halo_hits(engine).values()) == 2
    assert torch.equal(a, b)
    # accumulating dgrad: t += dgrad(gout), rounded like "store the dgrad, then add"
    gout = torch.randn(n, 96, device=DEV, generator=g).bfloat16()
    t = torch.randn(n, 96, device=DEV, generator=g).bfloat16()
    assert engine.lib().lgs_conv_dgrad_can_accumulate(km.h, 0, 96, 96, engine.LGS_BF16) == 1
    want = km.conv_dgrad(gout, w, False) + t
    got = km.conv_dgrad(gout, w, False, accumulate_into=t.clone())
    engine.tuning_set("HALO", 0)
    assert torch.equal(got, want)


def test_halo_tables_follow_the_knob_and_the_feature_dtype():
    """HALO = 1: the halo option follows the feature dtype (the fp32 parity path stays on k_conv_gather); default: nobody builds them"""
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, _, _ = make_batch([5], voxel=0.02, n_target=80000)
    conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, stride=1, dimension=3).to(DEV)
    with engine.tuning(HALO=1):
        engine.dispatch_counts(reset=True)
        conv(ME.SparseTensor(torch.randn(coords.shape[0], 32, device=DEV), torch.from_numpy(coords).to(DEV)))
        d = engine.dispatch_counts()
        assert not any(k.startswith("k_conv_halo") or k.startswith("k_build_halo") for k in d), d
        conv(ME.SparseTensor(torch.randn(coords.shape[0], 32, device=DEV).bfloat16(), torch.from_numpy(coords).to(DEV)))
        d = engine.dispatch_counts()
        assert any(k.startswith("k_conv_halo") for k in d) and any(k.startswith("k_build_halo") for k in d), d
    # and with the knob at its default (off) a bf16 tensor stays on k_conv_gather too
    assert engine.tuning_get("HALO") == 0
    engine.dispatch_counts(reset=True)
    conv(ME.SparseTensor(torch.randn(coords.shape[0], 32, device=DEV).bfloat16(), torch.from_numpy(coords).to(DEV)))
    assert not any(k.startswith("k_conv_halo") or k.startswith("k_build_halo") for k in engine.dispatch_counts())
