cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_rccl.py tests/test_gpu_ddp.py -m gpu -q -x -k "inline or block_fast or rccl or two_ranks" 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2; do python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f  single %.2f' % (d['ms_per_step'], d['single_scene']['ms_per_step']))"; done
LGS_WGRAD_INLINE_BELOW=0 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no inline: %.2f  single %.2f' % (d['ms_per_step'], d['single_scene']['ms_per_step']))"
