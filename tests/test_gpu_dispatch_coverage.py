"""No benchmarked kernel without a parity test.

Round 3 shipped a hot kernel (k_wgrad_wide: 19 % of the CLIP step's kernel time) that no test reached, because a size gate
(>= 200 k positions) sat between the test sizes and the benchmark size.  Every kernel launch of the engine is counted per launch
site -- kernel expression + template bindings, e.g. "k_wgrad_ps<KIND,NCS> [KIND=0,NCS=3]" or "k_conv_gather<T,2,3,4,1,...>
[T=unsignedshort]" -- and tests/conftest.py records the sites every GPU test hits.  This module runs LAST: it executes the steps
bench.py's default line times (BASELINE configs[1] bf16 + fp32, configs[2], configs[4] full / frozen, the per-rank path of N > 1
at a world of one rank, the single-scene line) on
the benchmark's 8-scene batch and asserts that each site they dispatch was also dispatched by an earlier (parity) test.

Reference workloads: /root/reference/scripts/train_models.sh (configs[1]), scripts/text_representation_train.sh:7 (configs[2]),
downstream/insseg (configs[4])."""
import argparse
import gc
import os
import sys

import pytest
import torch

import conftest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_SUITE = 120          # GPU tests that must have run before this one for the union to mean "the parity suite"


def _bench_sites():
    """-> {workload: set of launch sites of two training steps on the 8-scene benchmark batch}"""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(sync_bn=1, allreduce="ring", voxels=150000)
    out = {}

    def run(tag, workload, model_name, dtype, coords, feats, labels):
        ctx = bench.make_ctx(workload, model_name, coords, dev)
        model, ddp, opt = bench.make_trainer(model_name, dtype, dev, 1, args, ctx)
        engine.dispatch_counts(reset=True)
        for i in range(2):
            bench.train_step(model, ddp, opt, coords, feats, labels, dtype, 100 + i, ctx=ctx)
        torch.cuda.synchronize()
        out[tag] = set(engine.dispatch_counts(reset=True))
        del model, ddp, opt
        gc.collect()
        torch.cuda.empty_cache()

    c, f, l = (torch.from_numpy(a).to(dev) for a in make_batch(list(range(8)), voxel=0.02, n_target=150000))
    run("ce bf16 (configs[1], headline)", "ce", "Res16UNet34C", torch.bfloat16, c, f, l)
    run("ce bf16, balanced category sampling (scripts/train_models.sh:37)", "ce_balanced_sampled", "Res16UNet34C", torch.bfloat16, c, f, l)
    run("ce fp32 (parity path)", "ce", "Res16UNet34C", torch.float32, c, f, l)
    run("clip (configs[2])", "clip", "Res16UNet34D", torch.bfloat16, c, f, l)
    run("insseg full (configs[4])", "insseg", "InsSegRes16UNet34C", torch.bfloat16, c, f, l)
    run("insseg frozen trunk (configs[4])", "insseg_frozen", "InsSegRes16UNet34C", torch.bfloat16, c, f, l)
    # the per-rank code path of N > 1 (bench.py's `dp_path_world1` block): SyncBN through the engine's RCCL communicator + forced
    # bucket collectives with a world of one rank
    import torch.distributed as dist
    import MinkowskiEngine as ME
    from languagegroundedsemseg_amd.ddp import EngineComm
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % bench._free_port(), rank=0, world_size=1)
        try:
            ME.MinkowskiSyncBatchNorm.force_sync = True
            args.dp_world1 = True
            run("N > 1 per-rank path at a world of one rank", "ce", "Res16UNet34C", torch.bfloat16, c, f, l)
        finally:
            args.dp_world1 = False
            ME.MinkowskiSyncBatchNorm.force_sync = False
            EngineComm.close_all()
            dist.destroy_process_group()
    del c, f, l
    c, f, l = (torch.from_numpy(a).to(dev) for a in make_batch([1000], voxel=0.02, n_target=150000))
    run("single scene", "ce", "Res16UNet34C", torch.bfloat16, c, f, l)
    return out


def test_every_kernel_the_benchmark_dispatches_is_reached_by_a_parity_test(request):
    mine = request.node.nodeid
    suite = {k: v for k, v in conftest.DISPATCHED.items() if k != mine}
    if len(suite) < MIN_SUITE:
        pytest.skip("needs the whole `-m gpu` suite in the same session (%d GPU tests ran before this one, %d required)" % (len(suite), MIN_SUITE))
    # a launch site counts as covered only through a test that compared the HIP path with an INDEPENDENT reference: the CPU oracle,
    # a reference-generated golden fixture (both detected at run time by conftest), or a plain-torch restatement named by the
    # `parity` marker.  Bit-identity / property tests that compare the HIP path with itself do not cover anything.
    parity = {k: v for k, v in suite.items() if conftest.PARITY.get(k)}
    covered = set().union(*parity.values())
    self_only = set().union(*suite.values()) - covered
    bench_sites = _bench_sites()
    unreached = {}
    for tag, sites in bench_sites.items():
        assert len(sites) >= 10, (tag, sites)
        for s in sorted(sites - covered):
            unreached.setdefault(s, []).append(tag)
    for tag, sites in bench_sites.items():
        print("%-36s %3d launch sites" % (tag, len(sites)))
    print("GPU suite: %d tests, of which %d ran against an independent reference (%s); %d launch sites covered by those" % (
        len(suite), len(parity), ", ".join("%s: %d" % (a, sum(1 for v in conftest.PARITY.values() if v and a in v)) for a in ("oracle", "golden")),
        len(covered)))
    bench_all = set().union(*bench_sites.values())
    print("launch sites reached ONLY by self-comparison tests: %d, of which the benchmark dispatches %d" % (
        len(self_only), len(self_only & bench_all)))
    for s_ in sorted(self_only):
        print("   self-only%s  %s" % (" [BENCH]" if s_ in bench_all else "        ", s_))
    assert not unreached, "kernel launch sites the benchmark dispatches that NO parity test reaches:\n" + "\n".join(
        "  %s   <- %s" % (s, ", ".join(t)) for s, t in sorted(unreached.items()))


def test_dispatch_counters_see_the_gates():
    """the counters themselves: a 3^3 256 -> 256 weight gradient on a small map is k_wgrad_ps by default and k_wgrad_wide with
    the WW_MIN_ROWS gate lowered (the knob the teacher-forced 34D test uses)"""
    import MinkowskiEngine as ME
    from languagegroundedsemseg_amd import engine
    from helpers import small_scene
    coords = torch.from_numpy(small_scene(3, n=3000, extent=28)).to("cuda:0")
    x = ME.SparseTensor(torch.zeros(coords.shape[0], 3, device="cuda:0"), coords)
    km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
    n = x.F.shape[0]
    a = torch.randn(n, 256, device="cuda:0").bfloat16()
    g = torch.randn(n, 256, device="cuda:0").bfloat16()
    engine.dispatch_counts(reset=True)
    w0 = km.conv_wgrad(a, g, False)
    d0 = engine.dispatch_counts(reset=True)
    with engine.tuning(WW_MIN_ROWS=0):
        w1 = km.conv_wgrad(a, g, False)
    d1 = engine.dispatch_counts(reset=True)
    assert any(k.startswith("k_wgrad_ps") for k in d0) and not any(k.startswith("k_wgrad_wide") for k in d0), d0
    assert any(k.split()[0] == "k_wgrad_wide" for k in d1) and not any(k.startswith("k_wgrad_ps") for k in d1), d1
    assert engine.tuning_get("WW_MIN_ROWS") == 200000
    rel = float((w0 - w1).norm() / w0.norm())
    assert rel < 2e-3, rel                                    # two summation orders of the same bf16 products
