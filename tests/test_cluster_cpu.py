"""Pins oracle.pointgroup_clusters (KD-tree candidates + BFS) to a literal restatement of the reference's kernels on a
small case: the O(n^2) scan of bfs_cluster_kernel.cu:16-61 (d2 < r2 inside the batch segment, hits in ascending k) and
the queue BFS of bfs_cluster.cpp:54-101."""
from collections import deque

import numpy as np

from oracle import oracle as orc


def _literal(xyz, sem, radius, threshold, batch_offsets):
    n = xyz.shape[0]
    r2 = np.float32(radius) * np.float32(radius)
    nbr = []
    for b in range(len(batch_offsets) - 1):
        s, e = batch_offsets[b], batch_offsets[b + 1]
        for i in range(s, e):
            d = xyz[i] - xyz[s:e]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            nbr.append((np.nonzero(d2 < r2)[0] + s).tolist())        # includes i itself, like the kernel
    visited = np.zeros(n, bool)
    out = []
    for i in range(n):
        if visited[i]:
            continue
        cc = [i]; visited[i] = True
        q = deque([i])
        while q:
            cur = q.popleft()
            for j in nbr[cur]:
                if sem[j] != sem[cur] or visited[j]:
                    continue
                cc.append(j); visited[j] = True; q.append(j)
        if len(cc) >= threshold:
            out.append(cc)
    return out


def test_oracle_clusters_equal_literal_reference_restatement():
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 0.5, (500, 3)).astype(np.float32)
    b = rng.uniform(0, 0.5, (400, 3)).astype(np.float32)
    xyz = np.concatenate([a, b], 0)
    sem = rng.integers(0, 3, xyz.shape[0]).astype(np.int32)
    batch = np.concatenate([np.zeros(500, np.int32), np.ones(400, np.int32)])
    for radius, thr in ((0.05, 3), (0.08, 10)):
        ref = _literal(xyz, sem, radius, thr, [0, 500, 900])
        got = orc.pointgroup_clusters(xyz, sem, radius, thr, batch)
        assert len(ref) > 2 and ref == got          # same clusters, same order, same BFS order inside
