cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3o
timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | grep "3^3" | cut -c1-160
python bench.py --workload clip --no-cpu-baseline --no-single-scene --steps 6 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clip', d['ms_per_step'], d['phases']['stream_ms']); [print(t) for t in d['roofline']['discovery_step']['top_shapes']]; print(d['roofline']['discovery_step']['wgrad'])"
timeout 1800 python -m pytest tests/test_gpu_teacher_forced.py tests/test_gpu_parity_r2.py -m gpu -q -x -k "34d or 34D" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
