"""RCCL on the GPU box: the `nccl` backend (= RCCL on ROCm) initialised with a world of ONE rank,
BucketedDDP and MinkowskiSyncBatchNorm forced down their collective paths (dist.all_reduce on the flat gradient
buckets launched from the post-accumulate hooks, dist.all_gather_into_tensor / all_reduce of the SyncBN records).
A one-rank all-reduce is the identity, so the forced-collective step must reproduce the non-distributed step: this
executes the RCCL calls, their stream ordering against the engine's side stream, and the bucket views exactly as an
8-GPU run issues them."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_ddp_cpu import _free_port

pytestmark = pytest.mark.gpu


def _step(m, ddp, opt, coords, feats, dev, steps=2):
    """-> (gradients of the LAST step, parameters after it, gradients of the FIRST step)"""
    import MinkowskiEngine as ME
    out, first = None, None
    for _ in range(steps):
        ddp.zero_grad()
        logits, _ = m(ME.SparseTensor(feats, coords))
        logits.F.float().square().mean().backward()
        ddp.finalize()
        out = {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
        first = first or out
        opt.step()
    torch.cuda.synchronize()
    return out, {k: p.detach().float().cpu().clone() for k, p in m.named_parameters()}, first


def _worker(rank, port, ret):
    import torch.distributed as dist
    import MinkowskiEngine as ME
    from helpers import Cfg, deterministic_init
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    from languagegroundedsemseg_amd.models import load_model
    from languagegroundedsemseg_amd.synthetic import make_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        coords_np, feats_np, _ = make_batch([0, 1], voxel=0.05, n_target=6000)
        coords, feats = torch.from_numpy(coords_np).to(dev), torch.from_numpy(feats_np).to(dev).bfloat16()

        def build(sync):
            m = deterministic_init(load_model("Res16UNet14A")(3, 20, Cfg()), 42).to(dev).train()
            return ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(m) if sync else m
        # (1) gradient buckets through RCCL all_reduce vs no collective at all: bit-identical
        a = build(False)
        da = BucketedDDP(a, bucket_mb=1.0, force_collectives=True)
        assert da.reduce and len(da.buckets) > 3
        ga, pa, _ = _step(a, da, FlatSGD(da, lr=0.1, momentum=0.9, dampening=0.1, weight_decay=1e-4), coords, feats, dev)
        b = build(False)
        db = BucketedDDP(b, bucket_mb=1.0)
        assert not db.reduce
        gb, pb, _ = _step(b, db, FlatSGD(db, lr=0.1, momentum=0.9, dampening=0.1, weight_decay=1e-4), coords, feats, dev)
        same = all(torch.equal(ga[k], gb[k]) for k in gb) and all(torch.equal(pa[k], pb[k]) for k in pb) and set(ga) == set(gb)
        # (1b) the same buckets as reduce-scatter + all-gather IN PLACE on the flat bucket (reduce_scatter_tensor with the output
        # a slice of the input, all_gather_into_tensor back into it): RCCL's in-place forms, as `--allreduce rs_ag` issues them
        e = build(False)
        de = BucketedDDP(e, bucket_mb=1.0, force_collectives=True, allreduce="rs_ag")
        ge, pe, _ = _step(e, de, FlatSGD(de, lr=0.1, momentum=0.9, dampening=0.1, weight_decay=1e-4), coords, feats, dev)
        same_rs = all(torch.equal(ge[k], gb[k]) for k in gb) and all(torch.equal(pe[k], pb[k]) for k in pb) and set(ge) == set(gb)
        # (2) + SyncBN records through RCCL all_gather_into_tensor / all_reduce (Chan's combination of ONE record is
        # the same statistics up to the last float bit: compared with a tolerance, not bitwise)
        f32 = feats.float()
        d = build(False)
        dd = BucketedDDP(d, bucket_mb=1.0)
        _, _, gd = _step(d, dd, FlatSGD(dd, lr=1e-3), coords, f32, dev)
        ME.MinkowskiSyncBatchNorm.force_sync = True
        c = build(True)
        dc = BucketedDDP(c, bucket_mb=1.0, force_collectives=True)
        _, _, gc_ = _step(c, dc, FlatSGD(dc, lr=1e-3), coords, f32, dev)
        # (3) the whole-block autograd node with SyncBN inside (models._BasicBlockFunction calling ddp.sync_bn_forward /
        # sync_bn_backward) against the module-by-module path: same kernels, same collectives, same order -> bit-identical
        from languagegroundedsemseg_amd.me import block as _models
        calls = {"n": 0}
        orig_apply = _models._BasicBlockFunction.apply

        def counting_apply(*a, **k):
            calls["n"] += 1
            return orig_apply(*a, **k)
        _models._BasicBlockFunction.apply = counting_apply
        try:
            h = build(True)
            dh = BucketedDDP(h, bucket_mb=1.0, force_collectives=True)
            gh, ph, _ = _step(h, dh, FlatSGD(dh, lr=1e-3), coords, feats, dev)
            node_calls = calls["n"]
            _models._BLOCK_FUSED = False
            i = build(True)
            di = BucketedDDP(i, bucket_mb=1.0, force_collectives=True)
            gi, pi, _ = _step(i, di, FlatSGD(di, lr=1e-3), coords, feats, dev)
        finally:
            _models._BLOCK_FUSED = True
            _models._BasicBlockFunction.apply = orig_apply
        same_blk = (node_calls > 0 and calls["n"] == node_calls and set(gh) == set(gi) and all(torch.equal(gh[k], gi[k]) for k in gi)
                    and all(torch.equal(ph[k], pi[k]) for k in pi)
                    and all(torch.equal(a_.detach().cpu(), b_.detach().cpu()) for (_, a_), (_, b_) in zip(h.named_buffers(), i.named_buffers())))
        # (4) the collectives of (2) and (3) were issued by the ENGINE on its own RCCL communicator (one call per layer and
        # direction, csrc/lgs_comm.hip); the same step with torch.distributed's collectives between the split kernels: bit-identical
        from languagegroundedsemseg_amd.ddp import EngineComm
        used_engine_comm = any(v is not None for v in EngineComm._by_group.values())
        EngineComm.close_all()
        os.environ["LGS_SYNCBN_ENGINE_COMM"] = "0"
        try:
            j = build(True)
            dj = BucketedDDP(j, bucket_mb=1.0, force_collectives=True)
            gj, pj, _ = _step(j, dj, FlatSGD(dj, lr=1e-3), coords, feats, dev)
            torch_path = all(v is None for v in EngineComm._by_group.values())
        finally:
            del os.environ["LGS_SYNCBN_ENGINE_COMM"]
            EngineComm._by_group.clear()
        same_comm = (used_engine_comm and torch_path and set(gj) == set(gh) and all(torch.equal(gj[k], gh[k]) for k in gh)
                     and all(torch.equal(pj[k], ph[k]) for k in ph))
        ME.MinkowskiSyncBatchNorm.force_sync = False
        worst = max(float((gc_[k] - gd[k]).norm() / gd[k].norm().clamp_min(1e-12)) for k in gd)
        rm = float((c.bn0.bn.running_mean - d.bn0.bn.running_mean).abs().max())
        ret["out"] = (same, worst, rm, int(c.bn0.bn.num_batches_tracked), same_rs, same_blk, node_calls, same_comm)
        from languagegroundedsemseg_amd import engine as _engine
        ret["sites"] = sorted(_engine.dispatch_counts())      # launch sites this worker hit (for tests/test_gpu_dispatch_coverage.py)
    finally:
        from languagegroundedsemseg_amd.ddp import EngineComm as _EC
        _EC.close_all()
        dist.destroy_process_group()


def test_rccl_collective_paths_with_one_rank_reproduce_the_local_step():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), ret), nprocs=1, join=True)
    same, worst, rm, nbt, same_rs, same_blk, node_calls, same_comm = ret["out"]
    import conftest
    conftest.DISPATCHED["tests/test_gpu_rccl.py::worker"] = set(ret["sites"])
    assert same_comm, "SyncBN through the engine's RCCL communicator must equal SyncBN through torch.distributed's collectives bit for bit"
    assert same_blk, "SyncBN blocks as ONE autograd node (%d node calls) must equal the module-by-module SyncBN path bit for bit" % node_calls
    assert same, "RCCL all_reduce of the gradient buckets (world 1) must leave the step bit-identical"
    assert same_rs, "RCCL in-place reduce_scatter_tensor + all_gather_into_tensor of the buckets (world 1) must leave the step bit-identical"
    print("SyncBN over RCCL (world 1) vs local BatchNorm: worst gradient rel-L2 %.3e, running_mean diff %.3e" % (worst, rm))
    assert worst < 2e-2 and rm < 1e-5 and nbt == 2


def _own_group_worker(rank, port, ret):
    """a SECOND RCCL communicator (the process group SyncBN's exchanges get, knob SYNCBN_OWN_GROUP) next to the default group's
    bucket all-reduces, with a world of one rank: group creation, collectives on both, the result of the local step"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      LGS_SYNCBN_OWN_GROUP="2", LGS_SYNCBN_ENGINE_COMM="0")
    import torch.distributed as dist
    import MinkowskiEngine as ME
    from helpers import Cfg, deterministic_init
    from languagegroundedsemseg_amd import ddp as ddp_mod
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    from languagegroundedsemseg_amd.models import load_model
    from languagegroundedsemseg_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # device_id as bench.py passes it: the default group's communicator comes up eagerly and dist.new_group() SPLITS it
    # (ncclCommSplit) instead of initialising a second one from scratch -- the path the N > 1 bench takes
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        coords_np, feats_np, _ = make_batch([0, 1], voxel=0.05, n_target=6000)
        coords, feats = torch.from_numpy(coords_np).to(dev), torch.from_numpy(feats_np).to(dev)

        def run(sync):
            ME.MinkowskiSyncBatchNorm.force_sync = sync
            m = deterministic_init(load_model("Res16UNet14A")(3, 20, Cfg()), 42).to(dev).train()
            if sync:
                m = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(m)
            d = BucketedDDP(m, bucket_mb=1.0, force_collectives=sync)
            return _step(m, d, FlatSGD(d, lr=1e-3), coords, feats, dev)
        g1, p1, _ = run(True)
        groups = list(ddp_mod._OWN_GROUP.values())
        own = len(groups) == 1 and groups[0] is not dist.group.WORLD and dist.get_backend(groups[0]) == "nccl"
        g0, p0, _ = run(False)
        worst = max(float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-12)) for k in g0)
        ret["own"], ret["worst"], ret["keys"] = own, worst, set(g1) == set(g0)
    finally:
        ME.MinkowskiSyncBatchNorm.force_sync = False
        dist.destroy_process_group()


def test_syncbn_exchanges_on_their_own_rccl_group_reproduce_the_local_step():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_own_group_worker, args=(_free_port(), ret), nprocs=1, join=True)
    assert ret["own"], "SyncBN did not create / use its own process group"
    # one record combined by Chan's formula == the local statistics up to fp32 round-off, which flips a few ReLU gates (cf. the
    # tolerance of tests/test_gpu_ddp.py): measured 2.3e-3 on the worst tensor
    assert ret["keys"] and ret["worst"] < 2e-2, ret["worst"]
