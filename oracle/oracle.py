"""ctypes/numpy front-end of the CPU ORACLE (oracle/sparse_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing under ``languagegroundedsemseg_amd/`` or
``MinkowskiEngine/`` imports this module; it is used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the checker.

Parity status: *unpinned against MinkowskiEngine 0.5.4* (un-vendored, absent; see the
header of sparse_oracle.c) -- pinned against dense ``torch.nn.functional.conv3d`` /
``conv_transpose3d`` known answers (tests/test_oracle_dense.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """Compile sparse_oracle.c with gcc (needs no GPU)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "sparse_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        L.orc_unique_coords.restype = i64
        L.orc_unique_coords.argtypes = [vp, i64, vp, vp]
        L.orc_stride_coords.restype = i64
        L.orc_stride_coords.argtypes = [vp, i64, i32, vp, vp]
        L.orc_kernel_map.restype = i64
        L.orc_kernel_map.argtypes = [vp, i64, vp, i64, ctypes.c_int, i32, vp, vp, vp]
        L.orc_conv_forward.restype = None
        L.orc_conv_forward.argtypes = [vp, i64, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, i64, vp, i64]
        L.orc_conv_dgrad.restype = None
        L.orc_conv_dgrad.argtypes = [vp, i64, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, i64, vp, i64]
        L.orc_conv_wgrad.restype = None
        L.orc_conv_wgrad.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, i64, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _coords(c):
    c = np.ascontiguousarray(c, dtype=np.int32)
    assert c.ndim == 2 and c.shape[1] == 4, "coords must be [N,4] (batch,x,y,z)"
    return c


def unique_coords(coords):
    """-> (unique_index[nu] ascending, inverse[N]); first occurrence wins."""
    c = _coords(coords)
    n = c.shape[0]
    ui = np.empty(max(n, 1), np.int64)
    inv = np.empty(max(n, 1), np.int64)
    nu = lib().orc_unique_coords(_p(c), n, _p(ui), _p(inv))
    return ui[:nu].copy(), inv[:n].copy()


def stride_coords(coords, ts_out):
    """-> (out_coords[no,4], parent[N]): unique floor(c/ts_out)*ts_out, first-occurrence order."""
    c = _coords(coords)
    n = c.shape[0]
    oc = np.empty((max(n, 1), 4), np.int32)
    par = np.empty(max(n, 1), np.int64)
    no = lib().orc_stride_coords(_p(c), n, int(ts_out), _p(oc), _p(par))
    return oc[:no].copy(), par[:n].copy()


def kernel_map(in_coords, out_coords, ks, ts_in):
    """-> (k[M] int32, in_row[M] int64, out_row[M] int64) with c_in = c_out + off_k*ts_in."""
    ci, co = _coords(in_coords), _coords(out_coords)
    cap = max(co.shape[0] * ks ** 3, 1)
    kk = np.empty(cap, np.int32)
    ki = np.empty(cap, np.int64)
    ko = np.empty(cap, np.int64)
    m = lib().orc_kernel_map(_p(ci), ci.shape[0], _p(co), co.shape[0], int(ks), int(ts_in), _p(kk), _p(ki), _p(ko))
    return kk[:m].copy(), ki[:m].copy(), ko[:m].copy()


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _km(km):
    k, i, o = km
    return (np.ascontiguousarray(k, np.int32), np.ascontiguousarray(i, np.int64), np.ascontiguousarray(o, np.int64))


def conv_forward(x, w, km, n_out, bias=None):
    """x[n_in,Cin], w[K,Cin,Cout] -> out[n_out,Cout]."""
    x, w = _f32(x), _f32(w)
    if w.ndim == 2:
        w = w[None]
    k, i, o = _km(km)
    cin, cout = w.shape[1], w.shape[2]
    assert x.shape[1] == cin
    out = np.zeros((n_out, cout), np.float32)
    b = _f32(bias).reshape(-1) if bias is not None else None
    lib().orc_conv_forward(_p(x), x.shape[0], cin, _p(w), cout, _p(b) if b is not None else None,
                           _p(k), _p(i), _p(o), k.shape[0], _p(out), n_out)
    return out


def conv_dgrad(gout, w, km, n_in):
    gout, w = _f32(gout), _f32(w)
    if w.ndim == 2:
        w = w[None]
    k, i, o = _km(km)
    cin, cout = w.shape[1], w.shape[2]
    assert gout.shape[1] == cout
    gin = np.zeros((n_in, cin), np.float32)
    lib().orc_conv_dgrad(_p(gout), gout.shape[0], cout, _p(w), cin, _p(k), _p(i), _p(o), k.shape[0], _p(gin), n_in)
    return gin


def conv_wgrad(x, gout, km, K):
    x, gout = _f32(x), _f32(gout)
    k, i, o = _km(km)
    cin, cout = x.shape[1], gout.shape[1]
    gw = np.zeros((K, cin, cout), np.float32)
    lib().orc_conv_wgrad(_p(x), cin, _p(gout), cout, int(K), _p(k), _p(i), _p(o), k.shape[0], _p(gw))
    return gw


def transpose_map(km):
    """Map of the transposed convolution that undoes `km` (in/out swapped, same k)."""
    k, i, o = km
    return k, o, i


def canonical_order(coords):
    """Row permutation that sorts coords lexicographically by (b,x,y,z) -- for set-equality checks."""
    c = np.asarray(coords)
    return np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))


def quantize(points, affine=None, labels=None, ignore_label=-100, batch_index=0):
    """CPU restatement of the reference's voxelisation (test oracle for lgs_voxelize / lgs_label_vote):
         coords_aug = np.floor(hstack(coords, 1) @ rigid_transformation.T[:, :3])     lib/voxelizer.py:136-139
         _, unique_map = ME.utils.sparse_quantize(coords_aug, return_index=True)        lib/voxelizer.py:142
         sparse_quantize(coords, feats, labels, ignore_label=...)                      lib/voxelizer.py:284
    (dedup: first occurrence wins, indices ascending; a voxel whose points disagree on the label gets ignore_label).
    The affine product is evaluated elementwise in float64 in the fixed order ((x*a0 + y*a1) + z*a2) + a3.
    -> coords int32 [N,4] (batch column first), unique_index, inverse, voxel labels (or None)."""
    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    a = np.eye(4, dtype=np.float64)
    if affine is not None:
        m = np.asarray(affine, dtype=np.float64)
        a[:m.shape[0], :m.shape[1]] = m
    cols = []
    for r in range(3):
        v = ((p[:, 0] * a[r, 0] + p[:, 1] * a[r, 1]) + p[:, 2] * a[r, 2]) + a[r, 3]
        cols.append(np.floor(v).astype(np.int32))
    coords = np.stack([np.full(p.shape[0], batch_index, np.int32)] + cols, 1)
    ui, inv = unique_coords(coords)
    lab = None
    if labels is not None:
        ln = np.asarray(labels).astype(np.int64)
        lab = ln[ui].copy()
        mism = ln != lab[inv]
        lab[np.unique(inv[mism])] = ignore_label
    return coords, ui, inv, lab


def pointgroup_clusters(xyz, semantic_label, radius, threshold, batch_idx=None):
    """CPU restatement of PG_OP.ballquery_batch_p + PG_OP.bfs_cluster (test oracle for lgs_cluster):
         /root/reference/downstream/insseg/lib/bfs/ops/src/bfs_cluster_kernel.cu:16-61   d2 < radius^2 inside the batch segment
         /root/reference/downstream/insseg/lib/bfs/ops/src/bfs_cluster.cpp:54-101       BFS over same-label neighbours,
                                                                                         components visited by ascending start index,
                                                                                         kept if size >= threshold
    d2 = (ox-x)^2 + (oy-y)^2 + (oz-z)^2 in float32, left to right (numpy elementwise ops, no FMA).  Candidate pairs come
    from a KD-tree with a slightly larger radius and are then filtered with that exact float32 test.
    -> list of clusters, each a list of point indices in BFS order (cluster order = the reference's)."""
    from collections import deque
    from scipy.spatial import cKDTree
    p = np.asarray(xyz, dtype=np.float32)
    n = p.shape[0]
    sem = np.asarray(semantic_label)
    b = np.zeros(n, np.int64) if batch_idx is None else np.asarray(batch_idx)
    r2 = np.float32(radius) * np.float32(radius)
    pairs = cKDTree(p.astype(np.float64)).query_pairs(float(radius) * 1.001, output_type="ndarray")
    d = p[pairs[:, 0]] - p[pairs[:, 1]]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    ok = (d2 < r2) & (b[pairs[:, 0]] == b[pairs[:, 1]])
    pairs = pairs[ok]
    nbrs = [[] for _ in range(n)]
    for a, c in pairs:
        nbrs[a].append(c); nbrs[c].append(a)
    for l in nbrs:
        l.sort()                      # the ball query lists neighbours by ascending index (k = start..end)
    visited = np.zeros(n, bool)
    clusters = []
    for i in range(n):
        if visited[i]:
            continue
        cc = [i]; visited[i] = True
        q = deque([i])
        while q:
            cur = q.popleft()
            for j in nbrs[cur]:
                if sem[j] != sem[cur] or visited[j]:
                    continue
                cc.append(j); visited[j] = True; q.append(j)
        if len(cc) >= threshold:
            clusters.append(cc)
    return clusters
