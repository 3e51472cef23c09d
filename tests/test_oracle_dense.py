"""Pins the CPU oracle against an independent known-answer source: dense torch conv3d /
conv_transpose3d on a densified grid.  The reference holds no tests or
golden vectors for this path and MinkowskiEngine is absent, so this is the pin."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc

G = 12  # grid side


def rand_scene(seed, n=260, batches=2, lo=0, hi=G):
    rng = np.random.default_rng(seed)
    c = np.stack([rng.integers(0, batches, n), rng.integers(lo, hi, n), rng.integers(lo, hi, n),
                  rng.integers(lo, hi, n)], 1).astype(np.int32)
    ui, _ = orc.unique_coords(c)
    return c[ui]


def densify(coords, feats, B, C, shift=0, side=G):
    vol = torch.zeros(B, C, side, side, side, dtype=torch.float64)
    c = torch.from_numpy(coords.astype(np.int64))
    vol[c[:, 0], :, c[:, 1] + shift, c[:, 2] + shift, c[:, 3] + shift] = torch.from_numpy(feats).double()
    return vol


def dense_weight(w, ks):
    # w[K,Cin,Cout], k = a + ks*b + ks*ks*c (first spatial axis fastest) -> wd[Cout,Cin,a,b,c]
    K, cin, cout = w.shape
    wd = torch.from_numpy(w).double().reshape(ks, ks, ks, cin, cout)  # index [c][b][a]
    return wd.permute(4, 3, 2, 1, 0).contiguous()


def test_unique_first_occurrence():
    c = np.array([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1], [1, 1, 1, 1], [0, 2, 2, 2]], np.int32)
    ui, inv = orc.unique_coords(c)
    assert ui.tolist() == [0, 1, 3]
    assert inv.tolist() == [0, 1, 0, 2, 1]


def test_stride_floor_negative():
    c = np.array([[0, -1, 0, 3], [0, -2, 1, 2], [0, -3, 0, 0], [1, -1, 0, 3]], np.int32)
    oc, par = orc.stride_coords(c, 2)
    assert oc.tolist() == [[0, -2, 0, 2], [0, -4, 0, 0], [1, -2, 0, 2]]
    assert par.tolist() == [0, 0, 1, 2]


@pytest.mark.parametrize("seed", [0, 1])
def test_conv3_s1_matches_dense(seed):
    rng = np.random.default_rng(100 + seed)
    coords = rand_scene(seed)
    n, cin, cout = coords.shape[0], 5, 7
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = rng.standard_normal((27, cin, cout)).astype(np.float32)
    km = orc.kernel_map(coords, coords, 3, 1)
    y = orc.conv_forward(x, w, km, n)
    vol = densify(coords, x, 2, cin).requires_grad_(True)
    wd = dense_weight(w, 3).requires_grad_(True)
    yd = F.conv3d(vol, wd, padding=1)
    c = torch.from_numpy(coords.astype(np.int64))
    ys = yd[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]
    assert np.abs(ys.detach().numpy() - y).max() < 1e-4
    # backward: dgrad / wgrad against dense autograd
    g = rng.standard_normal((n, cout)).astype(np.float32)
    (ys * torch.from_numpy(g).double()).sum().backward()
    gin = orc.conv_dgrad(g, w, km, n)
    gind = vol.grad[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy()
    assert np.abs(gind - gin).max() < 1e-4
    gw = orc.conv_wgrad(x, g, km, 27)
    gwd = wd.grad.permute(4, 3, 2, 1, 0).reshape(27, cin, cout).numpy()  # back to [K,Cin,Cout]
    assert np.abs(gwd - gw).max() < 1e-3


def test_conv2_s2_and_transpose_match_dense():
    rng = np.random.default_rng(7)
    coords = rand_scene(3)
    n, cin, cout = coords.shape[0], 4, 6
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = rng.standard_normal((8, cin, cout)).astype(np.float32)
    oc, parent = orc.stride_coords(coords, 2)
    km = orc.kernel_map(coords, oc, 2, 1)
    assert km[0].shape[0] == n  # exactly one pair per fine voxel
    y = orc.conv_forward(x, w, km, oc.shape[0])
    vol = densify(coords, x, 2, cin)
    yd = F.conv3d(vol, dense_weight(w, 2), stride=2)
    c = torch.from_numpy(oc.astype(np.int64))
    ys = yd[c[:, 0], :, c[:, 1] // 2, c[:, 2] // 2, c[:, 3] // 2].numpy()
    assert np.abs(ys - y).max() < 1e-4
    # transposed conv: coarse -> fine on the cached fine map
    wt = rng.standard_normal((8, cout, cin)).astype(np.float32)  # [K, Cin_tr=cout, Cout_tr=cin]
    kmt = orc.transpose_map(km)
    z = orc.conv_forward(y, wt, kmt, n)
    vol_c = torch.zeros(2, cout, G // 2, G // 2, G // 2, dtype=torch.float64)
    vol_c[c[:, 0], :, c[:, 1] // 2, c[:, 2] // 2, c[:, 3] // 2] = torch.from_numpy(y).double()
    # conv_transpose3d weight [Cin, Cout, a,b,c]; out[2p + a] += in[p] * w[a]
    wtd = torch.from_numpy(wt).double().reshape(2, 2, 2, cout, cin).permute(3, 4, 2, 1, 0).contiguous()
    zd = F.conv_transpose3d(vol_c, wtd, stride=2)
    f = torch.from_numpy(coords.astype(np.int64))
    zs = zd[f[:, 0], :, f[:, 1], f[:, 2], f[:, 3]].numpy()
    assert np.abs(zs - z).max() < 1e-4


def test_negative_coords_and_shift_invariance():
    # a rigid integer shift (pl_BaselineTrainer.py:294) must not change a stride-1 conv's result
    rng = np.random.default_rng(11)
    coords = rand_scene(5)
    x = rng.standard_normal((coords.shape[0], 3)).astype(np.float32)
    w = rng.standard_normal((27, 3, 4)).astype(np.float32)
    y0 = orc.conv_forward(x, w, orc.kernel_map(coords, coords, 3, 1), coords.shape[0])
    sh = coords.copy()
    sh[:, 1:] += np.array([-37, 5, -101], np.int32)
    y1 = orc.conv_forward(x, w, orc.kernel_map(sh, sh, 3, 1), coords.shape[0])
    assert np.array_equal(y0, y1)


def test_empty_and_single():
    e = np.zeros((0, 4), np.int32)
    ui, inv = orc.unique_coords(e)
    assert ui.shape[0] == 0
    one = np.array([[0, 5, 5, 5]], np.int32)
    km = orc.kernel_map(one, one, 3, 1)
    assert km[0].tolist() == [13] and km[1].tolist() == [0]
