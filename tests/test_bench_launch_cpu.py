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
