"""Synthetic ScanNet200-shaped scenes (there is no network for the real dataset).

A procedurally generated room -- floor + 4 walls + 6-8 axis-aligned boxes (5 visible faces each) -- is
sampled as a surface point cloud (>= 4 points per voxel area, sigma = 1 mm noise), rotated by a random
yaw, voxelised with floor(p / voxel) and de-duplicated, exactly the layout the reference's collate
function hands to the model: coords int32 [N,4] = (batch, x, y, z), feats float32 [N,3] colours,
labels int [N] (lib/transforms.py:396-422).  The footprint is scaled so one scene has ~n_target voxels.
"""
import numpy as np


def _plane(rng, origin, u, v, density):
    """points on the parallelogram origin + a*u + b*v, a,b in [0,1), ~density points / m^2"""
    area = np.linalg.norm(np.cross(u, v))
    n = max(int(area * density), 1)
    ab = rng.random((n, 2))
    return origin[None] + ab[:, :1] * u[None] + ab[:, 1:] * v[None]


def make_scene(seed, voxel=0.02, n_target=150000, tol=0.05, num_labels=200, ignore_frac=0.10, ignore_label=-1):
    """-> (coords[N,3] int32 unique voxels, feats[N,3] float32 in [0,255], labels[N] int64)"""
    rng = np.random.default_rng(seed)
    scale = 1.0
    for _ in range(12):
        r = np.random.default_rng(seed)  # same layout every iteration, only the footprint changes
        L, W, H = 4.0 * scale, 3.0 * scale, 2.4
        density = 4.0 / (voxel * voxel)
        pts = [_plane(r, np.zeros(3), np.array([L, 0, 0.]), np.array([0, W, 0.]), density)]  # floor
        pts.append(_plane(r, np.zeros(3), np.array([L, 0, 0.]), np.array([0, 0, H]), density))
        pts.append(_plane(r, np.array([0, W, 0.]), np.array([L, 0, 0.]), np.array([0, 0, H]), density))
        pts.append(_plane(r, np.zeros(3), np.array([0, W, 0.]), np.array([0, 0, H]), density))
        pts.append(_plane(r, np.array([L, 0, 0.]), np.array([0, W, 0.]), np.array([0, 0, H]), density))
        nbox = int(r.integers(6, 9))
        for _b in range(nbox):
            sx, sy, sz = r.uniform(0.3, 1.2) * scale, r.uniform(0.3, 1.0) * scale, r.uniform(0.3, 1.5)
            ox, oy = r.uniform(0, max(L - sx, 1e-3)), r.uniform(0, max(W - sy, 1e-3))
            o = np.array([ox, oy, 0.0])
            ex, ey, ez = np.array([sx, 0, 0.]), np.array([0, sy, 0.]), np.array([0, 0, sz])
            pts.append(_plane(r, o + ez, ex, ey, density))        # top
            pts.append(_plane(r, o, ex, ez, density))             # 4 sides
            pts.append(_plane(r, o + ey, ex, ez, density))
            pts.append(_plane(r, o, ey, ez, density))
            pts.append(_plane(r, o + ex, ey, ez, density))
        p = np.concatenate(pts, 0)
        p += r.normal(0.0, 0.001, p.shape)
        yaw = r.uniform(0, 2 * np.pi)
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        p = (p - p.mean(0)) @ R.T                      # rigid augmentation about the origin -> negative coords
        vox = np.floor(p / voxel).astype(np.int32)
        vox = np.unique(vox, axis=0)
        n = vox.shape[0]
        if abs(n - n_target) <= tol * n_target:
            break
        scale *= np.sqrt(n_target / max(n, 1)) ** 0.9
    perm = rng.permutation(vox.shape[0])               # dataset order is arbitrary, not spatial
    vox = np.ascontiguousarray(vox[perm])
    feats = rng.uniform(0, 255, (vox.shape[0], 3)).astype(np.float32)
    labels = rng.integers(0, num_labels, vox.shape[0]).astype(np.int64)
    labels[rng.random(vox.shape[0]) < ignore_frac] = ignore_label
    return vox, feats, labels


def make_batch(seeds, voxel=0.02, n_target=150000, shift_seed=None, **kw):
    """Collate scenes like ME.utils.sparse_collate (batch index in column 0) and apply the trainer's
    per-step random integer shift in [0,100)^3 (pl_BaselineTrainer.py:294) when shift_seed is given.
    Colours are normalised like model_step: /255 - 0.5 (pl_BaselineTrainer.py:298-299)."""
    cs, fs, ls = [], [], []
    for b, s in enumerate(seeds):
        v, f, l = make_scene(s, voxel=voxel, n_target=n_target, **kw)
        cs.append(np.concatenate([np.full((v.shape[0], 1), b, np.int32), v], 1))
        fs.append(f / 255.0 - 0.5)
        ls.append(l)
    coords = np.concatenate(cs, 0)
    if shift_seed is not None:
        sh = (np.random.default_rng(shift_seed).random(3) * 100).astype(np.int32)
        coords[:, 1:] += sh[None]
    return coords, np.concatenate(fs, 0).astype(np.float32), np.concatenate(ls, 0)


def text_anchors(num_labels=200, dim=512, seed=1234):
    """stand-in for clip_feats_scannet_200.pkl (lib/datasets/prior_info.py:25-28): T ~ N(0,1)^{200 x 512}"""
    return np.random.default_rng(seed).standard_normal((num_labels, dim)).astype(np.float32)


def morton_order(coords):
    """permutation that sorts collated coords [N,4] = (batch, x, y, z) batch-major, then by the 3-D Morton code of the
    voxel (experiments on input row order; the engine accepts any order)"""
    c = coords[:, 1:].astype(np.int64)
    c = c - c.min(0, keepdims=True)
    key = np.zeros(c.shape[0], np.uint64)
    for bit in range(20):
        for a in range(3):
            key |= ((c[:, a].astype(np.uint64) >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + a)
    key |= coords[:, 0].astype(np.uint64) << np.uint64(60)
    return np.argsort(key, kind="stable")
