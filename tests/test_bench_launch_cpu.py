"""`python bench.py --gpus N` must produce N ranks on its own (round 3: --gpus was parsed and never read, so the driver's command
shape would have run ONE rank on an 8-GPU node).  CPU-side checks of the launcher logic: the decision table, the command
line, and a real self_launch of two gloo ranks of a stub script (the ranks of bench.py itself need GPUs).
Reference: /root/reference/main.py:192-195 (`gpus=N` handed to Lightning, which spawns one process per GPU)."""
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_gpus_flag_decides_between_launching_and_being_a_rank():
    import bench
    assert bench.resolve_world(1, {}) == ("rank", 1)
    assert bench.resolve_world(8, {}) == ("launch", 8)
    assert bench.resolve_world(8, {"WORLD_SIZE": "8", "RANK": "3"}) == ("rank", 8)
    assert bench.resolve_world(1, {"WORLD_SIZE": "1"}) == ("rank", 1)
    with pytest.raises(RuntimeError, match="WORLD_SIZE=2"):
        bench.resolve_world(8, {"WORLD_SIZE": "2"})
    with pytest.raises(RuntimeError):
        bench.resolve_world(1, {"WORLD_SIZE": "8"})


def test_launch_command_is_the_drivers_torchrun_shape():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_self_launch_runs_n_ranks_and_rank0_prints_one_line(tmp_path):
    import bench
    stub = tmp_path / "stub_rank.py"
    out = tmp_path / "line.json"
    stub.write_text(textwrap.dedent("""
        import json, os, sys
        import torch, torch.distributed as dist
        sys.path.insert(0, %r)
        import bench
        gpus = int(sys.argv[sys.argv.index("--gpus") + 1])
        mode, world = bench.resolve_world(gpus, os.environ)
        assert mode == "rank" and world == gpus
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            open(%r, "w").write(json.dumps({"n_gpus": world, "sum": float(t.item()), "self": os.environ.get("LGS_BENCH_SELF_LAUNCHED"),
                                            "addr": os.environ["MASTER_ADDR"]}))
        dist.barrier()
        dist.destroy_process_group()
    """ % (ROOT, str(out))))
    rc = bench.self_launch(2, ["--gpus", "2"], script=str(stub))
    assert rc == 0
    rec = json.loads(out.read_text())
    assert rec == {"n_gpus": 2, "sum": 3.0, "self": "1", "addr": "127.0.0.1"}


def test_self_launch_propagates_a_rank_failure(tmp_path):
    import bench
    stub = tmp_path / "bad_rank.py"
    stub.write_text("import sys; sys.exit(3)\n")
    assert bench.self_launch(2, [], script=str(stub)) != 0


def _full_record():
    """the biggest record a default run produces: round 5's committed line (21 KB: every secondary block) plus this round's keys"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    full["step_ms"] = {"median": 27.1234567, "p90": 27.9, "min": 26.9, "max": 31.2, "all": [27.1] * 20}
    full["warmup_ms"], full["warmup_extra"] = [40.0, 28.0, 27.5, 27.2, 27.1], 3
    rf = full["roofline"]
    rf.setdefault("hbm_frac", rf["frac"])
    rf.setdefault("arithmetic_intensity_flop_per_byte", 42.0)
    full["balanced"] = {"ms_per_step": 27.7, "sampled": {"ms_per_step": 28.3}}
    full["rccl_ranks"]["comm_create_s"] = None
    return full


def test_stdout_line_stays_under_4_kb_and_round_trips():
    """round 5's line had grown to 21 KB and the driver recorded `parsed: null`: the ONE stdout line carries the contract keys,
    `roofline`, `cpu_baseline` and one number per secondary workload; everything else goes to the details file"""
    import bench
    full = _full_record()
    text = bench.headline(full, "gpurun_out/bench_full.json")
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "details"):
        assert k in line, k
    assert line["config"]["workload"].startswith("Res16UNet34C") and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["step_ms"]["median"] == 27.12 and line["secondary"]["clip_ms"] > 100
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-6


def test_stdout_line_of_an_eight_rank_run_fits_too():
    import bench
    full = _full_record()
    full["n_gpus"] = 8
    full["per_rank"] = [dict(full["per_rank"][0], rank=r, ddp={"allreduce_exposed_wait_ms": 0.41234, "syncbn_collective_ms": 1.2345})
                        for r in range(8)]
    full["rccl_ranks"].update(comm_create_s=3.21, backend="nccl")
    text = bench.headline(full, None)
    assert len(text) < 4096
    line = json.loads(text)
    assert len(line["ranks"]["ms_per_step"]) == 8 and line["ranks"]["comm_create_s"] == 3.21


def test_warmup_settles_on_two_agreeing_rounds():
    import bench
    assert not bench._settled([[448.0, 31.5, 27.2, 26.9, 25.2]])                                      # one round says nothing
    assert not bench._settled([[448.0, 40.0, 35.0, 56.0, 50.0], [41.3, 35.5, 30.0, 27.0, 25.3]])      # round 5's driver run: still moving
    assert not bench._settled([[448.0, 31.5, 29.2, 28.9, 25.2], [31.1, 28.0, 26.9, 27.0, 25.3]])      # agreeing inside, 7 % below the round before
    assert bench._settled([[481.8, 31.8, 27.4, 27.2, 25.6], [31.8, 27.9, 27.3, 27.2, 25.7]])          # this round's measured warm-up
    assert bench._settled([[328.7, 31.7, 27.0, 26.0, 26.1, 25.9], [33.4, 27.9, 26.0, 26.1, 25.9]])    # a secondary block (second step still slow)
    s = bench.step_stats([27.0, 27.2, 27.1, 56.0])
    assert s["median"] == pytest.approx(27.15) and s["max"] == 56.0 and s["min"] == 27.0
