cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "large_map or wide or big" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-150
python bench.py --workload clip --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clip %.2f' % d['ms_per_step'], d['phases']['stream_ms'])"
