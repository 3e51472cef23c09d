"""Full-size checks (BASELINE configs[1]: 8 scenes, ~1.2 M voxels @2cm).  Round 5: the dominant launch shape (3^3 96 -> 96) and the
level 0 <-> 1 strided / transposed pair run against the ORACLE itself at this size (bottom of the file, ~2 minutes of host time);
a whole network would need many minutes, so the rest of the HIP path is held against identities any correct sparse convolution
satisfies:

* coordinate maps against numpy set arithmetic (dedup, stride-2 coarsening = unique(floor(c / 2) * 2));
* kernel-map pair counts: offset k and its mirror 26-k have the same number of pairs, the centre has N;
* conv with a one-hot weight (single offset, identity matrix) copies exactly the neighbour row a numpy hash lookup finds;
* adjointness <conv(x), g> = <x, dgrad(g)> = <W, wgrad(x, g)>: ties forward, dgrad and wgrad kernels together;
* linearity conv(a x + b y) = a conv(x) + b conv(y);
* bf16 and fp32 paths agree to bf16 precision on the same map.
"""
import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from languagegroundedsemseg_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def batch():
    coords, feats, labels = make_batch(list(range(8)), voxel=0.02, n_target=150000)
    return coords


def _key(c):
    c = c.astype(np.int64)
    return ((c[:, 0] << 48) | ((c[:, 1] + (1 << 15)) << 32) | ((c[:, 2] + (1 << 15)) << 16) | (c[:, 3] + (1 << 15)))


def test_fullsize_maps_against_numpy_sets(batch):
    coords = batch
    n = coords.shape[0]
    assert n > 1_000_000
    x = ME.SparseTensor(torch.zeros(n, 1, device=DEV), torch.from_numpy(coords).to(DEV))
    assert np.array_equal(np.sort(_key(x.C.cpu().numpy())), np.sort(_key(coords)))      # no voxel lost or invented
    mgr = x.coordinate_manager
    key = x.coordinate_map_key
    ref = coords
    for lvl in range(1, 5):
        key = mgr.stride(key, 2)
        ts = 2 ** lvl
        ref = np.concatenate([ref[:, :1], (ref[:, 1:] // ts) * ts], 1)
        want = np.unique(_key(ref))
        got = np.sort(_key(mgr.get_coordinates(key).cpu().numpy()))
        assert np.array_equal(got, want), "stride-%d map differs from numpy unique(floor(c/%d)*%d)" % (ts, ts, ts)


def test_fullsize_kernel_map_counts_are_mirror_symmetric(batch):
    coords = batch
    n = coords.shape[0]
    x = ME.SparseTensor(torch.zeros(n, 1, device=DEV), torch.from_numpy(coords).to(DEV))
    km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
    k, i, o = km.export()
    k = k.cpu().numpy()
    cnt = np.bincount(k, minlength=27)
    assert cnt[13] == n
    for kk in range(13):
        assert cnt[kk] == cnt[26 - kk], (kk, cnt[kk], cnt[26 - kk])
    # every pair is a real neighbour: c_in - c_out == offset_k
    ii, oo = i.cpu().numpy().astype(np.int64), o.cpu().numpy().astype(np.int64)
    C = x.C.cpu().numpy().astype(np.int64)
    d = C[ii, 1:] - C[oo, 1:]
    off = np.stack([k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1], 1)
    assert np.array_equal(d, off) and np.array_equal(C[ii, 0], C[oo, 0])


@pytest.mark.parametrize("k_sel", [0, 4, 13, 22])
def test_fullsize_one_hot_conv_copies_the_neighbour_row(batch, k_sel):
    coords = batch
    n, c = coords.shape[0], 32
    f = torch.randn(n, c, device=DEV)
    x = ME.SparseTensor(f, torch.from_numpy(coords).to(DEV))
    conv = ME.MinkowskiConvolution(c, c, kernel_size=3, dimension=3).to(DEV)
    with torch.no_grad():
        conv.kernel.zero_()
        conv.kernel[k_sel] = torch.eye(c, device=DEV)
    y = conv(x).F
    # numpy hash lookup of the neighbour at offset k_sel (first spatial axis fastest, ME's convention)
    off = np.array([k_sel % 3 - 1, (k_sel // 3) % 3 - 1, k_sel // 9 - 1], np.int64)
    C = x.C.cpu().numpy().astype(np.int64)
    keys = _key(C)
    order = np.argsort(keys)
    nb = C.copy()
    nb[:, 1:] += off
    q = _key(nb)
    pos = np.searchsorted(keys[order], q)
    pos = np.clip(pos, 0, n - 1)
    hit = keys[order][pos] == q
    src = order[pos]
    want = torch.zeros_like(f)
    hit_t = torch.from_numpy(hit).to(DEV)
    want[hit_t] = f[torch.from_numpy(src[hit]).to(DEV)]
    assert torch.equal(y, want)                               # fp32 MFMA with an identity weight is exact


def test_fullsize_adjoint_identity_ties_forward_dgrad_wgrad(batch):
    coords = batch
    n, cin, cout = coords.shape[0], 32, 64
    g = torch.Generator(device=DEV).manual_seed(0)
    x0 = torch.randn(n, cin, device=DEV, generator=g)
    go = torch.randn(n, cout, device=DEV, generator=g)
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 6e-3)):
        conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(DEV)
        xf = x0.detach().clone().to(dtype).requires_grad_(True)
        x = ME.SparseTensor(xf, torch.from_numpy(coords).to(DEV))
        y = conv(x).F
        gy = go.to(dtype)
        y.backward(gy)
        a = (y.double() * gy.double()).sum().item()
        b = (xf.detach().double() * xf.grad.double()).sum().item()
        c = (conv.kernel.detach().double() * conv.kernel.grad.double()).sum().item()
        scale = (y.double().norm() * gy.double().norm()).item()
        assert abs(a - b) / scale < tol, (dtype, a, b)
        assert abs(a - c) / scale < tol, (dtype, a, c)


def test_fullsize_linearity_and_bf16_vs_fp32(batch):
    coords = batch
    n, cin, cout = coords.shape[0], 96, 96
    g = torch.Generator(device=DEV).manual_seed(1)
    xa = torch.randn(n, cin, device=DEV, generator=g)
    xb = torch.randn(n, cin, device=DEV, generator=g)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(DEV)
    c = torch.from_numpy(coords).to(DEV)
    with torch.no_grad():
        sa = ME.SparseTensor(xa, c)
        mk = dict(coordinate_map_key=sa.coordinate_map_key, coordinate_manager=sa.coordinate_manager)
        ya = conv(sa).F
        yb = conv(ME.SparseTensor(xb, **mk)).F
        yc = conv(ME.SparseTensor(0.5 * xa - 2.0 * xb, **mk)).F
        ref = 0.5 * ya - 2.0 * yb
        assert (yc - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
        y16 = conv(ME.SparseTensor(xa.bfloat16(), **mk)).F.float()
        ya_r = conv(ME.SparseTensor(xa.bfloat16().float(), **mk)).F      # same rounded inputs, fp32 path
        err = (y16 - ya_r).norm().item() / ya_r.norm().item()
        assert err < 6e-3, err                                            # bf16 weights + bf16 output rounding


def _neighbour_rows(C, k_sel):
    """numpy hash lookup: (hit mask, source row) of the neighbour at offset k_sel of every voxel (first spatial axis fastest)"""
    off = np.array([k_sel % 3 - 1, (k_sel // 3) % 3 - 1, k_sel // 9 - 1], np.int64)
    keys = _key(C)
    order = np.argsort(keys)
    nb = C.copy()
    nb[:, 1:] += off
    q = _key(nb)
    pos = np.clip(np.searchsorted(keys[order], q), 0, len(keys) - 1)
    return keys[order][pos] == q, order[pos]


def _dispatched(before, after, name):
    return sum(v - before.get(k, 0) for k, v in after.items() if name in k)


@pytest.mark.parametrize("k_sel", [2, 13, 21])
def test_fullsize_wide_one_hot_conv_copies_the_neighbour_row(batch, k_sel):
    """configs[2] width at configs[1] size: the 2-D blocked wide kernel (>= 256 output channels, bf16) with a one-hot weight is an
    exact row copy -- every gathered row, every 64-channel stage and every output column tile lands where numpy says"""
    from languagegroundedsemseg_amd import engine
    coords = batch
    n, c = coords.shape[0], 512
    g = torch.Generator(device=DEV).manual_seed(k_sel)
    f = torch.randn(n, c, device=DEV, generator=g).to(torch.bfloat16)
    x = ME.SparseTensor(f, torch.from_numpy(coords).to(DEV))
    conv = ME.MinkowskiConvolution(c, c, kernel_size=3, dimension=3).to(DEV)
    with torch.no_grad():
        conv.kernel.zero_()
        conv.kernel[k_sel] = torch.eye(c, device=DEV)
    before = engine.dispatch_counts()
    with torch.no_grad():
        y = conv(x).F
    assert _dispatched(before, engine.dispatch_counts(), "k_conv_wide") >= 1, "the wide kernel was not the one that ran"
    hit, src = _neighbour_rows(x.C.cpu().numpy().astype(np.int64), k_sel)
    want = torch.zeros_like(f)
    hit_t = torch.from_numpy(hit).to(DEV)
    want[hit_t] = f[torch.from_numpy(src[hit]).to(DEV)]
    assert torch.equal(y, want)


def test_fullsize_wide_adjoint_identity_ties_forward_dgrad_wgrad(batch):
    """<conv(x), g> = <x, dgrad(g)> = <W, wgrad(x, g)> at 1.2 M voxels x 512 -> 512 channels (bf16): k_conv_wide forward, k_conv_wide
    on the mirrored weights and k_wgrad_wide (compacted pair lists) describe the same bilinear form"""
    from languagegroundedsemseg_amd import engine
    coords = batch
    n, cin, cout = coords.shape[0], 512, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3).to(DEV)
    xf = torch.randn(n, cin, device=DEV, generator=g).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(n, cout, device=DEV, generator=g).to(torch.bfloat16)
    x = ME.SparseTensor(xf, torch.from_numpy(coords).to(DEV))
    before = engine.dispatch_counts()
    y = conv(x).F
    y.backward(gy)
    torch.cuda.synchronize()
    after = engine.dispatch_counts()
    assert _dispatched(before, after, "k_conv_wide") >= 2 and _dispatched(before, after, "k_wgrad_wide") >= 1
    a = (y.double() * gy.double()).sum().item()
    b = (xf.detach().double() * xf.grad.double()).sum().item()
    c = (conv.kernel.detach().double() * conv.kernel.grad.double()).sum().item()
    scale = (y.double().norm() * gy.double().norm()).item()
    assert abs(a - b) / scale < 6e-3, (a, b)
    assert abs(a - c) / scale < 6e-3, (a, c)


# ------------------------------------------------------------------------------------------- round 5: the dominant launch shape
# itself, at the benchmark's size, against the ORACLE (not a property): level-0 3^3 96 -> 96 on the 1.2 M-voxel batch -- the launch
# `roofline` is quoted on.  The oracle's BLAS restatement (gather -> GEMM -> scatter per offset, oracle/backend.py "torch") needs about
# a minute on the host cores for forward + both gradients of this one layer, which is why the other full-size checks are properties.
_ORACLE_L0 = {}


def _oracle_l0(coords, x_np, w_np, g_np):
    if "v" not in _ORACLE_L0:
        from oracle.backend import OracleBackend
        prev = ME.set_backend(OracleBackend("torch"))
        try:
            conv = ME.MinkowskiConvolution(96, 96, kernel_size=3, dimension=3)
            with torch.no_grad():
                conv.kernel.copy_(torch.from_numpy(w_np))
            xf = torch.from_numpy(x_np).clone().requires_grad_(True)
            xs = ME.SparseTensor(xf, torch.from_numpy(coords))
            y = conv(xs).F
            y.backward(torch.from_numpy(g_np))
            _ORACLE_L0["v"] = (y.detach().numpy(), xf.grad.numpy(), conv.kernel.grad.numpy())
        finally:
            ME.set_backend(prev)
    return _ORACLE_L0["v"]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)], ids=["fp32", "bf16"])
def test_fullsize_dominant_launch_shape_against_the_oracle(batch, dtype, tol):
    """conv forward, dgrad and weight gradient of 3^3 96 -> 96 on all 1 205 389 voxels (13.7 M pairs) vs the oracle: relative L2 of
    each tensor (fp32: the split-operand path on the bf16 matrix pipe; bf16: inputs rounded to bf16 first, so only the kernels'
    own accumulation / output rounding is compared).  Row order: level-0 rows are the caller's rows on both sides."""
    coords = batch
    n = coords.shape[0]
    rng = np.random.default_rng(5)
    x_np = rng.standard_normal((n, 96), dtype=np.float32)
    g_np = rng.standard_normal((n, 96), dtype=np.float32)
    w_np = (rng.standard_normal((27, 96, 96), dtype=np.float32) / np.sqrt(27 * 96)).astype(np.float32)
    if dtype == torch.bfloat16:       # both sides see bf16-representable inputs
        x_np = torch.from_numpy(x_np).bfloat16().float().numpy()
        g_np = torch.from_numpy(g_np).bfloat16().float().numpy()
        key = "bf16"
    else:
        key = "fp32"
    if _ORACLE_L0.get("key") != key:
        _ORACLE_L0.clear()
        _ORACLE_L0["key"] = key
    oy, ogx, ogw = _oracle_l0(coords, x_np, w_np, g_np)
    conv = ME.MinkowskiConvolution(96, 96, kernel_size=3, dimension=3).to(DEV)
    with torch.no_grad():
        conv.kernel.copy_(torch.from_numpy(w_np))
    xf = torch.from_numpy(x_np).to(DEV).to(dtype).requires_grad_(True)
    xs = ME.SparseTensor(xf, torch.from_numpy(coords).to(DEV))
    y = conv(xs).F
    y.backward(torch.from_numpy(g_np).to(DEV).to(dtype))

    def rel(a, b):
        a, b = a.detach().double().cpu(), torch.from_numpy(b).double()
        return float((a - b).norm() / b.norm())
    e = (rel(y, oy), rel(xf.grad, ogx), rel(conv.kernel.grad, ogw))
    print("full size 96->96 %s: rel-L2 forward %.2e, dgrad %.2e, wgrad %.2e" % (key, *e))
    assert max(e) < tol, e


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 2e-2)], ids=["fp32", "bf16"])
def test_fullsize_strided_and_transposed_convolutions_against_the_oracle(batch, dtype, tol):
    """the level 0 <-> level 1 pair of the benchmark batch (conv1p1s2 / convtr7p2s2 shapes: 2^3 stride 2 down, transposed 2^3 up,
    res16unet.py:205,262) on all 1.2 M voxels vs the oracle: output (back on the input's rows), input gradient and both weight
    gradients; the oracle builds its own coarse map (rows compared by coordinate)"""
    from test_gpu_engine import run_both, rel_err
    coords = batch
    feats = np.random.default_rng(9).standard_normal((coords.shape[0], 32)).astype(np.float32)
    if dtype == torch.bfloat16:
        feats = torch.from_numpy(feats).bfloat16().float().numpy()

    def build():
        return [ME.MinkowskiConvolution(32, 64, kernel_size=2, stride=2, dimension=3),
                ME.MinkowskiConvolutionTranspose(64, 96, kernel_size=2, stride=2, dimension=3)]
    (h_out, h_g), (o_out, o_g) = run_both(build, coords, feats, dtype=dtype, oracle_impl="torch")
    e = [rel_err(h_out, o_out)] + [rel_err(a, b) for a, b in zip(h_g, o_g)]
    print("full size 2^3 s2 down + transposed up, %s: rel err out %.2e, d input %.2e, dW down %.2e, dW up %.2e" % ((str(dtype),) + tuple(e)))
    assert max(e) < tol, e
