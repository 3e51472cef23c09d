"""Shared test helpers (CPU + GPU suites)."""
import zlib

import numpy as np
import torch


def deterministic_init(model, seed=42):
    """Fill every parameter / running stat from a generator keyed by its NAME, so that two independently
    constructed models (the reference's files vs the build's models.py) get identical weights without
    shipping a state dict."""
    with torch.no_grad():
        for name, p in sorted(model.state_dict().items()):
            if not p.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7fffffff)
            if name.endswith("running_var"):
                v = torch.rand(p.shape, generator=g) * 0.5 + 0.75
            elif name.endswith("running_mean"):
                v = (torch.rand(p.shape, generator=g) - 0.5) * 0.1
            elif name.endswith("bn.weight"):
                v = torch.rand(p.shape, generator=g) * 0.5 + 0.75
            elif name.endswith("bias"):
                v = (torch.rand(p.shape, generator=g) - 0.5) * 0.2
            else:
                fan_in = p.shape[-2] * (p.shape[0] if p.dim() == 3 else 1) if p.dim() >= 2 else p.numel()
                a = (3.0 / max(fan_in, 1)) ** 0.5
                v = (torch.rand(p.shape, generator=g) * 2 - 1) * a
            p.copy_(v.to(p.dtype))
    return model


class Cfg:
    """the subset of config/config.py defaults the models read"""
    bn_momentum = 0.02
    conv1_kernel_size = 3
    dilations = [1, 1, 1, 1]


def small_scene(seed, n=1500, extent=24, batches=2):
    """random surface-ish voxels: a few noisy planes per scene, int32 [N,4]"""
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batches):
        pts = []
        for _ in range(3):
            axis = rng.integers(0, 3)
            p = rng.integers(0, extent, (n // (3 * batches) + 1, 3))
            p[:, axis] = rng.integers(0, extent) + rng.integers(0, 2, p.shape[0])
            pts.append(p)
        p = np.unique(np.concatenate(pts, 0), axis=0)
        p = p[rng.permutation(p.shape[0])] - extent // 2
        out.append(np.concatenate([np.full((p.shape[0], 1), b), p], 1))
    return np.concatenate(out, 0).astype(np.int32)


def canon(coords):
    c = np.asarray(coords)
    return np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))


def triples_as_set(coords_in, coords_out, k, i, o):
    """kernel map as a set of (k, in_coord, out_coord) tuples -- row-order independent"""
    ci, co = np.asarray(coords_in), np.asarray(coords_out)
    k, i, o = np.asarray(k), np.asarray(i), np.asarray(o)
    rec = np.concatenate([k[:, None].astype(np.int64), ci[i].astype(np.int64), co[o].astype(np.int64)], 1)
    return rec[np.lexsort(rec.T[::-1])]
