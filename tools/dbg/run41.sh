cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r3.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2 3; do python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], 'single %.2f' % d['single_scene']['ms_per_step'])"; done
