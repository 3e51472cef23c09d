"""Voxelisation oracle (oracle.quantize) against numpy restating the reference call sequence verbatim in form:
np.floor(hstack(coords, 1) @ rigid.T[:, :3]) (lib/voxelizer.py:136-139) + ME.utils.sparse_quantize semantics (first
occurrence wins, ascending indices, label collision -> ignore_label), and the host path of ME.utils.sparse_quantize."""
import numpy as np
import torch

import MinkowskiEngine as ME
from oracle import oracle as orc


def _points(seed, n=4000):
    rng = np.random.default_rng(seed)
    base = rng.integers(-40, 40, (n // 3, 3)).astype(np.float64) * 0.05
    pts = np.concatenate([base + rng.uniform(0.003, 0.047, base.shape) for _ in range(3)], 0)   # 3 points per cell on average
    pts = pts[rng.permutation(pts.shape[0])].astype(np.float32)
    labels = rng.integers(0, 5, pts.shape[0]).astype(np.int64)
    return pts, labels


def _rigid(seed, voxel=0.05):
    rng = np.random.default_rng(seed)
    th = rng.uniform(-np.pi, np.pi)
    R = np.eye(4)
    R[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    V = np.eye(4)
    V[:3, :3] /= voxel
    M = R @ V
    M[:3, 3] = rng.uniform(-3, 3, 3)
    return M


def test_oracle_quantize_equals_reference_formulation():
    pts, labels = _points(0)
    M = _rigid(1)
    coords, ui, inv, lab = orc.quantize(pts, M, labels, ignore_label=-1, batch_index=2)
    homo = np.hstack((pts.astype(np.float64), np.ones((pts.shape[0], 1))))
    ref = np.floor(homo @ M.T[:, :3])                        # BLAS dot product: may differ in the last ulp ...
    near = np.abs(ref - np.floor(homo @ M.T[:, :3] + 1e-9)) > 0      # ... which only matters exactly on a cell boundary
    assert near.sum() == 0
    assert np.array_equal(coords[:, 1:], ref.astype(np.int32)) and np.all(coords[:, 0] == 2)
    # ME.utils.sparse_quantize semantics on those integer coordinates (numpy restatement)
    _, first, inverse = np.unique(coords, axis=0, return_index=True, return_inverse=True)
    assert np.array_equal(ui, np.sort(first))
    assert np.array_equal(coords[ui][inv], coords)
    for v in np.random.default_rng(3).integers(0, ui.shape[0], 200):
        members = labels[inv == v]
        assert lab[v] == (members[0] if np.all(members == members[0]) else -1)
        assert lab[v] == -1 or lab[v] == labels[ui[v]]


def test_host_sparse_quantize_matches_oracle():
    pts, labels = _points(5)
    q = 0.05
    M = np.eye(4)
    M[:3, :3] /= q
    coords, ui, inv, lab = orc.quantize(pts, M, labels, ignore_label=-100)
    c, f, l, idx, inverse = ME.utils.sparse_quantize(torch.from_numpy(pts), torch.from_numpy(pts), torch.from_numpy(labels),
                                                     ignore_label=-100, return_index=True, return_inverse=True, quantization_size=q)
    # float32 division (host path) vs float64 product (oracle): identical away from cell boundaries, which _points avoids
    assert np.array_equal(idx.numpy(), ui) and np.array_equal(inverse.numpy(), inv)
    assert np.array_equal(c.numpy(), coords[ui][:, 1:]) and np.array_equal(l.numpy(), lab)
