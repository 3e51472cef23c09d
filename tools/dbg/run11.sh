cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3l
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_parity_r3.py tests/test_gpu_ddp.py tests/test_gpu_syncbn.py tests/test_gpu_rccl.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['phases'], d['single_scene'])"
PYTHONPATH=$GRAFT_REPO_ROOT python tools/hosttime.py 2>&1 | head -5
