"""ME.utils.sparse_quantize / sparse_collate -- the CPU data-side helpers the reference's dataset code
calls in DataLoader workers (lib/voxelizer.py:142, lib/datasets/scannet.py:238,328, lib/dataset.py:300,
lib/transforms.py:421,497-498).  numpy/torch on the host by design: they run before the batch exists."""
import collections.abc

import numpy as np
import torch


def _floor_int(coords):
    if isinstance(coords, torch.Tensor):
        return torch.floor(coords).to(torch.int32) if coords.is_floating_point() else coords.to(torch.int32)
    coords = np.asarray(coords)
    return np.floor(coords).astype(np.int32) if np.issubdtype(coords.dtype, np.floating) else coords.astype(np.int32)


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False, return_inverse=False,
                    return_maps_only=False, quantization_size=None, device="cpu"):
    """floor, dedup; first occurrence wins, surviving indices ascending.  If `labels` is given, voxels whose
    points disagree on the label get `ignore_label`."""
    is_torch = isinstance(coordinates, torch.Tensor)
    c = coordinates
    if quantization_size is not None:
        c = c / quantization_size
    c = _floor_int(c)
    cn = c.cpu().numpy() if is_torch else c
    assert cn.ndim == 2
    _, first, inverse = np.unique(cn, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)  # ascending original index
    unique_map = first[order]
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    inverse_map = rank[np.asarray(inverse).reshape(-1)]
    if labels is not None:
        ln = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
        lab = ln[unique_map].copy()
        # label collision inside a voxel -> ignore
        mism = ln != lab[inverse_map]
        if mism.any():
            lab[np.unique(inverse_map[mism])] = ignore_label
    conv = (lambda a: torch.from_numpy(np.ascontiguousarray(a))) if is_torch else (lambda a: a)
    if return_maps_only:
        return (conv(unique_map), conv(inverse_map)) if return_inverse else conv(unique_map)
    out = [conv(cn[unique_map])]
    if features is not None:
        out.append(features[conv(unique_map)] if is_torch and isinstance(features, torch.Tensor) else np.asarray(features)[unique_map])
    if labels is not None:
        out.append(conv(lab))
    if return_index:
        out.append(conv(unique_map))
    if return_inverse:
        out.append(conv(inverse_map))
    return out[0] if len(out) == 1 else tuple(out)


def batched_coordinates(coords, dtype=torch.int32, device=None):
    assert isinstance(coords, collections.abc.Sequence)
    D = coords[0].shape[1]
    n = sum(int(c.shape[0]) for c in coords)
    out = torch.zeros((n, D + 1), dtype=dtype)
    s = 0
    for b, c in enumerate(coords):
        c = torch.as_tensor(np.asarray(c) if not isinstance(c, torch.Tensor) else c)
        c = torch.floor(c).to(dtype) if c.is_floating_point() else c.to(dtype)
        e = s + c.shape[0]
        out[s:e, 0] = b
        out[s:e, 1:] = c
        s = e
    return out.to(device) if device is not None else out


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """-> (bcoords[N,1+D] with batch index in column 0, feats, labels)."""
    bcoords = batched_coordinates(coords, dtype=dtype, device=device)

    def _cat(xs):
        ts = [x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x)) for x in xs]
        return torch.cat(ts, 0)

    f = _cat(feats)
    if labels is not None:
        return bcoords, f, _cat(labels)
    return bcoords, f


class SparseCollation:
    def __init__(self, limit_numpoints=-1, dtype=torch.int32, device=None):
        self.limit_numpoints, self.dtype, self.device = limit_numpoints, dtype, device

    def __call__(self, list_data):
        coords, feats, labels = list(zip(*list_data))
        return sparse_collate(coords, feats, labels, dtype=self.dtype, device=self.device)
