cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -x -k "block_fast" > $O/pytest_blk.log 2>&1; tail -15 $O/pytest_blk.log
for f in 0 1; do echo "== LGS_BLOCK_FUSED=$f"; LGS_BLOCK_FUSED=$f python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['stream_ms'], d['phases']['host_enqueue_ms']['forward'], d['phases']['host_enqueue_ms']['backward'], d['single_scene']['ms_per_step'])"; done > $O/bench_ab.txt 2>&1; cat $O/bench_ab.txt
HOSTTIME_SCENES=1 python tools/hosttime.py 2>&1 | grep "host enqueue" 
