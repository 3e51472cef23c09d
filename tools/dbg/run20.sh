cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3u
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "block_fast or ddp or bucket or trajectory or strided or wgrad" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
HOSTTIME_SCENES=1 python tools/hosttime.py 2>&1 | grep "host enqueue" 
python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['stream_ms'], d['phases']['host_enqueue_ms']['forward'], d['phases']['host_enqueue_ms']['backward'], d['single_scene']['ms_per_step'])"
