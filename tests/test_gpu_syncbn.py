"""SyncBN on the engine's split kernels (lgs_bn_stats -> all-gather -> lgs_bn_sync_combine -> lgs_bn_apply;
lgs_bn_backward_reduce -> all-reduce -> lgs_bn_backward_apply with the device-side 1/N), run by two gloo ranks that share
the one GPU of the box, against single-process full-batch BatchNorm (main.py:122-123)."""
import pytest
import torch
import torch.nn as nn

from test_ddp_cpu import run_distributed

pytestmark = pytest.mark.gpu
N, C = 5000, 96


def _data(dtype):
    torch.manual_seed(11)
    full = (torch.randn(N, C) * 2 + 0.5).to(dtype).float()
    res = torch.randn(N, C).to(dtype).float()
    g = torch.randn(N, C).to(dtype).float()
    return full, res, g


def _job(rank, world, dtype_name, relu, use_res, ipc=False, layers=1):
    import os
    if ipc:
        os.environ["LGS_SYNCBN_IPC"] = "1"                      # (read when languagegroundedsemseg_amd.tuning.host() is asked)
        os.environ["LGS_MBOX_ALLOW_COARSE"] = "1"               # both ranks share ONE device: a coarse-grained mailbox is coherent here
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import MinkowskiEngine as ME
    from languagegroundedsemseg_amd.ddp import sync_batch_norm
    dtype = getattr(torch, dtype_name)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    full, res, g = _data(dtype)
    sl = slice(0, 1800) if rank == 0 else slice(1800, N)          # ragged shards
    x = full[sl].to(dev).to(dtype).requires_grad_(True)
    r = res[sl].to(dev).to(dtype).requires_grad_(True)
    mod = ME.MinkowskiSyncBatchNorm(C, momentum=0.02).to(dev)
    with torch.no_grad():
        mod.bn.weight.copy_(torch.linspace(0.5, 1.5, C)); mod.bn.bias.copy_(torch.linspace(-0.2, 0.2, C))
    y = sync_batch_norm(x, mod.bn, residual=r if use_res else None, relu=relu)
    y.backward(g[sl].to(dev).to(dtype))
    torch.cuda.synchronize()
    used = None
    if ipc:
        from languagegroundedsemseg_amd import engine
        from languagegroundedsemseg_amd.ddp import EngineComm
        used = [c is not None and c.ipc for c in EngineComm._by_group.values()]
        sites = engine.dispatch_counts()
        assert used == [True] and sites.get("k_mbox_allgather", 0) == 1 and sites.get("k_mbox_allreduce", 0) == 1, (used, sites)
        # more exchanges than the ring has slots, of different widths, the ranks drifting apart in between
        for i in range(layers):
            cc = (32, 256, 64, 128, 96, 512, 8)[i % 7]
            xi = (torch.randn(700 + 300 * rank, cc, device=dev) * (1 + i) + rank).to(dtype)
            mi = ME.MinkowskiSyncBatchNorm(cc).to(dev)
            yi = sync_batch_norm(xi.requires_grad_(True), mi.bn, relu=bool(i % 2))
            if rank == i % 2:
                torch.cuda.synchronize()                          # one rank waits, the other runs ahead into the next exchange
            yi.float().square().mean().backward()
            st = torch.cat([mi.bn.running_mean, mi.bn.running_var]).cpu()
            box = [None, None]
            import torch.distributed as dist
            dist.all_gather_object(box, st)
            assert torch.equal(box[0], box[1]), "layer %d: the ranks combined different statistics" % i
        EngineComm.close_all()
    return (y.detach().float().cpu(), x.grad.float().cpu(), r.grad.float().cpu() if use_res else None,
            mod.bn.running_mean.cpu(), mod.bn.running_var.cpu(), mod.bn.weight.grad.cpu(), mod.bn.bias.grad.cpu(),
            int(mod.bn.num_batches_tracked))


@pytest.mark.parity("plain torch BatchNorm on the concatenated batch")
@pytest.mark.parametrize("dtype_name,tol", [("float32", 2e-5), ("bfloat16", 2e-2)])
@pytest.mark.parametrize("relu,use_res", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("ipc", [False, True], ids=["collectives", "mailbox"])
def test_fused_sync_bn_two_ranks_match_full_batch(dtype_name, tol, relu, use_res, ipc):
    """ipc = True: the same exchange through the device-side mailboxes of csrc/lgs_comm.hip (knob SYNCBN_IPC): each rank's
    buffer mapped into the other process with hipIpc, one kernel per exchange that stores into both mailboxes and spins on
    the arrival flags -- here with both processes on ONE GPU, which exercises the ring / sequence / combination logic but not the
    visibility of peer stores across xGMI; followed by nine more exchanges (> ring depth) with the ranks out of step"""
    import functools
    r0, r1 = run_distributed(functools.partial(_job, dtype_name=dtype_name, relu=relu, use_res=use_res, ipc=ipc, layers=9 if ipc else 0))
    dtype = getattr(torch, dtype_name)
    full, res, g = _data(dtype)
    xf = full.clone().requires_grad_(True)
    rf = res.clone().requires_grad_(True)
    bn = nn.BatchNorm1d(C, momentum=0.02)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.2, 0.2, C))
    y = bn(xf)
    if use_res:
        y = y + rf
    if relu:
        y = torch.relu(y)
    y.backward(g)

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))
    assert rel(torch.cat([r0[0], r1[0]]), y.detach()) < tol
    assert rel(torch.cat([r0[1], r1[1]]), xf.grad) < 5 * tol
    if use_res:
        assert rel(torch.cat([r0[2], r1[2]]), rf.grad) < tol
    assert rel(r0[3], bn.running_mean) < max(tol, 1e-4) and rel(r1[4], bn.running_var) < max(tol, 1e-4)
    assert rel(r0[5] + r1[5], bn.weight.grad) < 5 * tol          # parameter grads stay local: their sum is the full one
    assert rel(r0[6] + r1[6], bn.bias.grad) < 5 * tol
    assert r0[7] == 1 and r1[7] == 1


def _diverging_job(rank, world):
    """rank 1 stops taking part after the first exchange; rank 0's next exchange must give up after LGS_MBOX_TIMEOUT_S and the call
    after that must fail with a message (advisor, round 5: the mailbox kernel used to spin for ever)"""
    import os
    import time
    os.environ.update(LGS_SYNCBN_IPC="1", LGS_MBOX_ALLOW_COARSE="1", LGS_MBOX_TIMEOUT_S="1.5", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import MinkowskiEngine as ME
    import torch.distributed as dist
    from languagegroundedsemseg_amd.ddp import EngineComm, sync_batch_norm
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    mod = ME.MinkowskiSyncBatchNorm(32).to(dev)
    x = torch.randn(500 + 100 * rank, 32, device=dev)
    with torch.no_grad():
        sync_batch_norm(x, mod.bn)                     # both ranks: communicator comes up, one exchange
    torch.cuda.synchronize()
    out = None
    if rank == 0:
        t0 = time.perf_counter()
        with torch.no_grad():
            sync_batch_norm(x, mod.bn)                 # nobody answers: the kernel waits 1.5 s, reports, returns
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
        try:
            with torch.no_grad():
                sync_batch_norm(x, mod.bn)
            out = ("no error", waited)
        except RuntimeError as e:
            out = (str(e), waited)
    dist.barrier()
    EngineComm.close_all()
    return out


def test_mailbox_exchange_gives_up_when_the_ranks_diverge():
    r0, r1 = run_distributed(_diverging_job)
    assert r1 is None
    msg, waited = r0
    assert "timed out" in msg and "diverged" in msg, msg
    assert 1.0 < waited < 20.0, waited
