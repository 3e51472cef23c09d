"""N > 1 path on CPU: world_size-2 gloo processes.  Covers the bucketed gradient all-reduce (overlapped with
backward through post-accumulate hooks), unused-parameter flushing (the reference runs DDP with
find_unused_parameters=True, main.py:193) and the packed SyncBatchNorm statistics exchange (main.py:122-123)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run_distributed(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.unused = nn.Linear(3, 3)       # never touched in forward

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _ddp_job(rank, world):
    from languagegroundedsemseg_amd.ddp import BucketedDDP
    torch.manual_seed(100 + rank)            # different init per rank: the wrapper must broadcast rank 0's
    net = Net()
    ddp = BucketedDDP(net, bucket_mb=0.0005)  # tiny buckets -> several async all-reduces
    assert len(ddp.buckets) > 1
    torch.manual_seed(7)
    full = torch.randn(8, 8)
    target = torch.randn(8, 4)
    x, y = full[rank * 4:(rank + 1) * 4], target[rank * 4:(rank + 1) * 4]
    for step in range(2):
        ddp.zero_grad()
        loss = ((ddp(x) - y) ** 2).mean()
        loss.backward()
        ddp.finalize()
    # parameters that received no gradient keep .grad = None (the optimiser skips them on every rank alike)
    grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    return grads, params


def test_bucketed_allreduce_equals_full_batch_gradient():
    (g0, p0), (g1, p1) = run_distributed(_ddp_job)
    assert torch.equal(p0, p1), "parameters must be broadcast from rank 0"
    assert torch.allclose(g0, g1, atol=1e-7), "gradients must be identical on all ranks"
    # reference: single process on the concatenated batch with rank 0's weights
    torch.manual_seed(100)
    net = Net()
    torch.manual_seed(7)
    full, target = torch.randn(8, 8), torch.randn(8, 4)
    loss = ((net(full) - target) ** 2).mean()
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    assert torch.allclose(g0, ref, atol=1e-6)


def _syncbn_job(rank, world):
    import MinkowskiEngine as ME
    from languagegroundedsemseg_amd.ddp import sync_batch_norm
    torch.manual_seed(3)
    full = torch.randn(10, 6) * 2 + 1
    x = full[:4] if rank == 0 else full[4:]            # ragged shards: 4 and 6 rows
    x = x.clone().requires_grad_(True)
    mod = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(nn.Sequential(ME.MinkowskiBatchNorm(6, momentum=0.02)))
    assert isinstance(mod[0], ME.MinkowskiSyncBatchNorm)
    bn = mod[0].bn
    y = sync_batch_norm(x, bn)
    w = torch.arange(1, 7, dtype=torch.float32)
    (y * w).sum().backward()
    return y.detach(), x.grad, bn.running_mean.clone(), bn.running_var.clone(), bn.weight.grad.clone()


def test_sync_batch_norm_matches_full_batch():
    r0, r1 = run_distributed(_syncbn_job)
    torch.manual_seed(3)
    full = (torch.randn(10, 6) * 2 + 1).requires_grad_(True)
    bn = nn.BatchNorm1d(6, momentum=0.02)
    y = bn(full)
    w = torch.arange(1, 7, dtype=torch.float32)
    (y * w).sum().backward()
    assert torch.allclose(torch.cat([r0[0], r1[0]]), y.detach(), atol=1e-5)
    assert torch.allclose(torch.cat([r0[1], r1[1]]), full.grad, atol=1e-5)
    assert torch.allclose(r0[2], bn.running_mean, atol=1e-6) and torch.allclose(r1[3], bn.running_var, atol=1e-6)
    # parameter grads stay local per rank (DDP averages them afterwards): their sum is the full-batch gradient
    assert torch.allclose(r0[4] + r1[4], bn.weight.grad, atol=1e-5)


def test_flat_sgd_equals_torch_sgd():
    """FlatSGD over the gradient buckets == torch.optim.SGD(momentum, dampening, weight_decay) of lib/solvers.py"""
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    torch.manual_seed(0)
    a, b = Net(), Net()
    b.load_state_dict(a.state_dict())
    ddp = BucketedDDP(a, bucket_mb=0.0005)
    fo = FlatSGD(ddp, lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    to = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    x, y = torch.randn(16, 8), torch.randn(16, 4)
    for step in range(4):
        ddp.zero_grad()
        ((a(x) - y) ** 2).mean().backward()
        ddp.finalize()
        fo.step()
        to.zero_grad(set_to_none=True)
        ((b(x) - y) ** 2).mean().backward()
        to.step()
    for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(p1, p2, atol=1e-6), n1
    assert torch.equal(a.unused.weight, b.unused.weight)   # untouched on both sides


def _multi_step_job(rank, world):
    """three optimiser steps (the second with 2-micro-batch accumulation under no_sync): every step's gradients must
    reach the flat buckets -- gradients autograd produced outside a slot on step 1 must still be collected on steps 2+"""
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    torch.manual_seed(100)
    net = Net()
    ddp = BucketedDDP(net, bucket_mb=0.0005)
    opt = FlatSGD(ddp, lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    torch.manual_seed(7)
    full, target = torch.randn(3, 8, 8), torch.randn(3, 8, 4)
    for step in range(3):
        x, y = full[step, rank * 4:(rank + 1) * 4], target[step, rank * 4:(rank + 1) * 4]
        ddp.zero_grad()
        if step == 1:                                   # accumulate two half micro-batches, reduce on the second
            with ddp.no_sync():
                (((ddp(x[:2]) - y[:2]) ** 2).mean() * 0.5).backward()
                ddp.finalize()
            (((ddp(x[2:]) - y[2:]) ** 2).mean() * 0.5).backward()
        else:
            ((ddp(x) - y) ** 2).mean().backward()
        ddp.finalize()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()])


def test_multi_step_and_accumulation_match_torch_sgd_on_the_full_batch():
    p0, p1 = run_distributed(_multi_step_job)
    assert torch.equal(p0, p1)
    torch.manual_seed(100)
    net = Net()
    to = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    torch.manual_seed(7)
    full, target = torch.randn(3, 8, 8), torch.randn(3, 8, 4)
    for step in range(3):
        to.zero_grad(set_to_none=True)
        ((net(full[step]) - target[step]) ** 2).mean().backward()
        to.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(p0, ref, atol=1e-6)


def _double_backward_job(rank, world):
    from languagegroundedsemseg_amd.ddp import BucketedDDP
    torch.manual_seed(0)
    net = Net()
    ddp = BucketedDDP(net, bucket_mb=0.0005)
    x = torch.randn(4, 8)
    ddp.zero_grad()
    ddp(x).sum().backward()
    try:
        ddp(x).sum().backward()
    except RuntimeError as e:
        ddp.finalize()
        return "after its bucket was reduced" in str(e)
    ddp.finalize()
    return False


def test_hook_after_reduce_raises():
    """a second backward without zero_grad()/no_sync() would accumulate into a bucket whose all-reduce may be in flight"""
    assert all(run_distributed(_double_backward_job))


def _rs_ag_job(rank, world):
    """the reduce-scatter + all-gather form of the bucket reduction (BucketedDDP(allreduce="rs_ag")): same gradients as
    the ring all-reduce, buckets padded to a multiple of the world size"""
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    torch.manual_seed(100)
    net = Net()
    ddp = BucketedDDP(net, bucket_mb=0.0005, allreduce="rs_ag")
    assert all(b["flat"].numel() % (4 * world) == 0 for b in ddp.buckets)
    opt = FlatSGD(ddp, lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    torch.manual_seed(7)
    full, target = torch.randn(3, 8, 8), torch.randn(3, 8, 4)
    for step in range(3):
        x, y = full[step, rank * 4:(rank + 1) * 4], target[step, rank * 4:(rank + 1) * 4]
        ddp.zero_grad()
        ((ddp(x) - y) ** 2).mean().backward()
        ddp.finalize()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()])


def test_reduce_scatter_all_gather_buckets_match_torch_sgd_on_the_full_batch():
    p0, p1 = run_distributed(_rs_ag_job)
    assert torch.equal(p0, p1)
    torch.manual_seed(100)
    net = Net()
    to = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    torch.manual_seed(7)
    full, target = torch.randn(3, 8, 8), torch.randn(3, 8, 4)
    for step in range(3):
        to.zero_grad(set_to_none=True)
        ((net(full[step]) - target[step]) ** 2).mean().backward()
        to.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(p0, ref, atol=1e-6)


class BranchNet(nn.Module):
    """a data-dependent branch: rank 0 never uses `b`, rank 1 never uses `c` (why the reference runs
    find_unused_parameters=True, main.py:193)"""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 8)
        self.b = nn.Linear(8, 4)
        self.c = nn.Linear(8, 4)
        self.d = nn.Linear(8, 8)

    def forward(self, x, use_b):
        h = torch.relu(self.a(self.d(x)))
        return self.b(h) if use_b else self.c(h)


def _uneven_unused_job(rank, world):
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    torch.manual_seed(5)
    net = BranchNet()
    ddp = BucketedDDP(net, bucket_mb=0.0001)          # one bucket per parameter or two: hooks fire in different orders per rank
    opt = FlatSGD(ddp, lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    assert len(ddp.buckets) >= 4
    torch.manual_seed(9)
    xs = torch.randn(2, 2, 4, 8)
    for step in range(2):
        ddp.zero_grad()
        ddp.module(xs[step, rank], use_b=(rank == 1)).pow(2).mean().backward()
        ddp.finalize()
        opt.step()
    # the reduced bucket slots (a parameter unused on THIS rank keeps .grad = None, its slot still holds the average)
    slots = torch.cat([b["flat"][:b["grad_elems"]] for b in ddp.buckets])
    return slots, torch.cat([p.detach().reshape(-1) for p in net.parameters()])


def test_ranks_with_different_unused_parameters_reduce_in_the_same_bucket_order():
    """collectives must be issued in a fixed bucket order on every rank: with per-rank unused sets a hook-order launch
    would pair bucket k of one rank with bucket j of the other (hang, or silently mixed gradients)"""
    (g0, p0), (g1, p1) = run_distributed(_uneven_unused_job)
    assert torch.allclose(g0, g1, atol=1e-7)
    # a parameter used on only ONE rank moves identically on both (the all-reduced "used" flag drives FlatSGD's mask)
    assert torch.equal(p0, p1)
    torch.manual_seed(5)
    net = BranchNet()
    to = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    torch.manual_seed(9)
    xs = torch.randn(2, 2, 4, 8)
    for step in range(2):
        to.zero_grad(set_to_none=True)
        (0.5 * (net(xs[step, 0], False).pow(2).mean() + net(xs[step, 1], True).pow(2).mean())).backward()
        to.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(p0, ref, atol=1e-6)


def _syncbn_own_group_job(rank, world):
    """LGS_SYNCBN_OWN_GROUP=2: the statistics exchanges of MinkowskiSyncBatchNorm run on a process group of their own (created
    collectively at the first exchange), next to all-reduces on the default group"""
    os.environ["LGS_SYNCBN_OWN_GROUP"] = "2"
    from languagegroundedsemseg_amd import ddp
    out = _syncbn_job(rank, world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)                                   # the default group still works beside it
    groups = list(ddp._OWN_GROUP.values())
    return out, float(t), len(groups), groups[0] is not dist.group.WORLD and dist.get_world_size(groups[0]) == world


def test_sync_batch_norm_on_its_own_process_group_matches_the_default_group():
    a = run_distributed(_syncbn_job)
    b = run_distributed(_syncbn_own_group_job)
    for r in range(2):
        out, s, n_groups, ok = b[r]
        assert s == 3.0 and n_groups == 1 and ok
        for x, y in zip(a[r], out):
            assert torch.equal(x, y) if torch.is_tensor(x) else x == y
