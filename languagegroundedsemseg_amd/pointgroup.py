"""PointGroup proposal clustering on the MI355X engine.

Mirror of /root/reference/downstream/insseg/lib/bfs/bfs.py: `ballquery_batch_p` + `bfs_cluster` (PG_OP CUDA ball query +
CPU BFS) are one fused device op here (`lgs_cluster`: radius-cell grid + union-find), `Clustering` keeps the reference's
constructor, `cluster_` / `cluster` / `get_instances` and their return formats.  Points of a cluster are listed in
ascending index order (the reference lists them in BFS order; its callers only scatter them into masks), clusters come in
the reference's order (ascending smallest member)."""
import ctypes

import numpy as np
import torch

from . import engine


def cluster_points(coords, semantic_label, radius, threshold, batch_idxs=None):
    """coords [N,3] float (HIP tensor), semantic_label [N] int -> (cluster_idxs int32 [M,2] = (cluster id, point index),
    cluster_offsets int32 [nCluster+1]), the output format of PG_OP.bfs_cluster (bfs_cluster.cpp:104-125)."""
    if not (isinstance(coords, torch.Tensor) and coords.is_cuda):
        raise RuntimeError("cluster_points runs on the MI355X engine: coords must be a HIP tensor (no CPU fallback)")
    L = engine.lib()
    dev = coords.device
    xyz = coords.detach().to(torch.float32).contiguous()
    sem = semantic_label.to(dev).to(torch.int32).contiguous()
    n = xyz.shape[0]
    bi = batch_idxs.to(dev).to(torch.int32).contiguous() if batch_idxs is not None else None
    vp = ctypes.c_void_p
    with torch.cuda.device(dev):
        comp = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(L.lgs_cluster_workspace_bytes(n)), 256), dtype=torch.uint8, device=dev)
        nc = ctypes.c_int32(0)
        engine.check(L.lgs_cluster(vp(xyz.data_ptr()), vp(bi.data_ptr()) if bi is not None else vp(None), vp(sem.data_ptr()), n,
                                   float(radius), int(threshold), vp(comp.data_ptr()), ctypes.byref(nc), vp(ws.data_ptr()),
                                   vp(torch.cuda.current_stream(dev).cuda_stream)))
    comp = comp[:n]
    keep = torch.nonzero(comp >= 0).flatten()
    if keep.numel() == 0:
        return torch.zeros((0, 2), dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    rep = comp[keep].long()
    order = torch.argsort(rep * n + keep, stable=True)          # clusters by representative, members ascending
    keep, rep = keep[order], rep[order]
    reps, counts = torch.unique_consecutive(rep, return_counts=True)
    cid = torch.repeat_interleave(torch.arange(reps.numel(), device=dev), counts)
    idxs = torch.stack([cid.int(), keep.int()], 1)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(counts, 0)]).int()
    assert reps.numel() == nc.value
    return idxs, offsets


class Clustering:
    """bfs.py:85-157 with the fused device op (vertices may be a numpy array or a tensor)."""

    def __init__(self, ignored_labels, class_mapping, thresh=0.03, closed_points=300, min_points=50, propose_points=100,
                 score_func=torch.max, device="cuda"):
        self.ignored_labels, self.thresh, self.closed_points = ignored_labels, thresh, closed_points
        self.min_points, self.propose_points, self.score_func = min_points, propose_points, score_func
        self.device = torch.device(device)
        self.class_mapping = class_mapping.to(self.device)

    def cluster_(self, vertices, labels):
        labels = labels.to(self.device)
        mask = torch.ones_like(labels, dtype=torch.bool)
        for ig in self.ignored_labels:
            mask &= self.class_mapping[labels] != ig
        object_idxs = mask.nonzero().view(-1)
        v = torch.as_tensor(np.asarray(vertices) if not isinstance(vertices, torch.Tensor) else vertices).to(self.device)
        v = v[object_idxs].float()
        if v.numel() == 0:
            return torch.zeros((0, 2), dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
        idx, off = cluster_points(v, labels[object_idxs], self.thresh, self.min_points)
        idx[:, 1] = object_idxs[idx[:, 1].long()].int()
        return idx, off

    def cluster(self, vertices, scores):
        labels = torch.max(scores, 1)[1].to(self.device)
        proposals_idx, proposals_offset = self.cluster_(vertices, labels)
        n = scores.shape[0]
        pred = torch.zeros((proposals_offset.shape[0] - 1, n), dtype=torch.int, device=proposals_idx.device)
        pred[proposals_idx[:, 0].long(), proposals_idx[:, 1].long()] = 1
        lab = labels[proposals_idx[:, 1][proposals_offset[:-1].long()].long()]
        keep = pred.sum(1) > self.propose_points
        return pred[keep], lab[keep]

    def get_instances(self, vertices, scores):
        pred, labels = self.cluster(vertices, scores)
        scores = scores.to(pred.device)
        out = {}
        for i in range(len(pred)):
            sc = self.score_func(scores[pred[i].bool(), labels[i]])
            out[i] = {"conf": sc.cpu().numpy(), "label_id": self.class_mapping.cpu()[labels[i]], "pred_mask": pred[i].cpu().numpy()}
        return out
