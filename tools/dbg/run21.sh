cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3v
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_engine.py tests/test_abi.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['stream_ms'], d['phases']['host_enqueue_ms']['forward'], d['phases']['host_enqueue_ms']['backward'], d['single_scene']['ms_per_step'])"
LGS_BLOCK_FUSED=0 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['stream_ms'], d['phases']['host_enqueue_ms']['forward'], d['phases']['host_enqueue_ms']['backward'], d['single_scene']['ms_per_step'])"
