"""ME.utils.sparse_quantize / sparse_collate -- the data-side helpers the reference's dataset code calls in DataLoader
workers (lib/voxelizer.py:142, lib/datasets/scannet.py:238,328, lib/dataset.py:300, lib/transforms.py:421,497-498).

Host inputs (numpy / CPU tensors) take the numpy path below, as in ME.  DEVICE tensors are quantised on the MI355X
(SURVEY 8f-1): `voxelize` = the reference's rigid transform + floor (lib/voxelizer.py:136-139) as one kernel,
dedup = the engine's coordinate insert (first occurrence wins, indices ascending), label collisions = lgs_label_vote."""
import collections.abc

import numpy as np
import torch


def _floor_int(coords):
    if isinstance(coords, torch.Tensor):
        return torch.floor(coords).to(torch.int32) if coords.is_floating_point() else coords.to(torch.int32)
    coords = np.asarray(coords)
    return np.floor(coords).astype(np.int32) if np.issubdtype(coords.dtype, np.floating) else coords.astype(np.int32)


def voxelize(points, affine=None, batch_index=0, quantization_size=None):
    """Device voxelisation: points [N,3] float (HIP tensor) -> int32 [N,4] = (batch, floor(A (x,y,z,1))).
    `affine`: 3x4 / 4x4 array-like (the reference's `rigid_transformation`, lib/voxelizer.py:129-139); default =
    identity scaled by 1/quantization_size."""
    import ctypes
    from .. import engine
    if not (isinstance(points, torch.Tensor) and points.is_cuda):
        raise RuntimeError("ME.utils.voxelize runs on the MI355X engine: points must be a HIP tensor")
    a = np.eye(4, dtype=np.float64)
    if affine is not None:
        m = np.asarray(affine.cpu() if isinstance(affine, torch.Tensor) else affine, dtype=np.float64)
        a[:m.shape[0], :m.shape[1]] = m
    elif quantization_size is not None:
        a[:3, :3] /= float(quantization_size)
    a12 = (ctypes.c_double * 12)(*a[:3, :4].reshape(-1).tolist())
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    assert pts.dim() == 2 and pts.shape[1] == 3
    with torch.cuda.device(pts.device):
        out = torch.empty((n, 4), dtype=torch.int32, device=pts.device)
        engine.check(engine.lib().lgs_voxelize(ctypes.c_void_p(pts.data_ptr()), n, a12, int(batch_index),
                                               ctypes.c_void_p(out.data_ptr()),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def _sparse_quantize_device(coordinates, features, labels, ignore_label, return_index, return_inverse, return_maps_only,
                            quantization_size):
    """sparse_quantize on HIP tensors: floor (+ 1/quantization_size) -> engine insert -> label vote, all on the device"""
    import ctypes
    from .. import engine
    from .core import get_backend
    c = coordinates
    if c.is_floating_point():
        if c.shape[1] == 3:
            q = voxelize(c, quantization_size=quantization_size)
        else:   # leading batch column: keep it, voxelise the spatial part
            q = voxelize(c[:, 1:].contiguous(), quantization_size=quantization_size)
            q[:, 0] = c[:, 0].to(torch.int32)
    else:
        assert quantization_size is None, "integer coordinates are already quantised"
        q = c.to(torch.int32)
        if q.shape[1] == 3:
            q = torch.cat([torch.zeros_like(q[:, :1]), q], 1)
    drop_batch = coordinates.shape[1] == 3
    mgr = get_backend().new_manager(q.device)
    _, nu, unique_map, inverse_map = mgr.insert(q.contiguous())
    lab = None
    if labels is not None:
        ln = labels.to(q.device).to(torch.int64).contiguous()
        lab = torch.empty(nu, dtype=torch.int64, device=q.device)
        sp = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        engine.check(engine.lib().lgs_label_vote(ctypes.c_void_p(ln.data_ptr()), int(ln.shape[0]), ctypes.c_void_p(unique_map.data_ptr()),
                                                 ctypes.c_void_p(inverse_map.data_ptr()), int(nu), int(ignore_label),
                                                 ctypes.c_void_p(lab.data_ptr()), sp))
        lab = lab.to(labels.dtype)
    if return_maps_only:
        return (unique_map, inverse_map) if return_inverse else unique_map
    qc = q.index_select(0, unique_map)
    out = [qc[:, 1:].contiguous() if drop_batch else qc]
    if features is not None:
        out.append(features.to(q.device).index_select(0, unique_map))
    if labels is not None:
        out.append(lab)
    if return_index:
        out.append(unique_map)
    if return_inverse:
        out.append(inverse_map)
    return out[0] if len(out) == 1 else tuple(out)


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False, return_inverse=False,
                    return_maps_only=False, quantization_size=None, device="cpu"):
    """floor, dedup; first occurrence wins, surviving indices ascending.  If `labels` is given, voxels whose
    points disagree on the label get `ignore_label`.  HIP tensors are processed on the device."""
    if isinstance(coordinates, torch.Tensor) and coordinates.is_cuda:
        return _sparse_quantize_device(coordinates, features, labels, ignore_label, return_index, return_inverse,
                                       return_maps_only, quantization_size)
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
