cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 400 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('400 steps: %.2f ms, single %.2f' % (d['ms_per_step'], d['single_scene']['ms_per_step']))"
timeout 300 python bench.py --workload clip --steps 40 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clip 40 steps: %.2f ms' % d['ms_per_step'])"
timeout 300 python bench.py --scenes 1 --steps 300 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 scene 300 steps: %.2f ms' % d['ms_per_step'])"
timeout 300 python bench.py --scenes 2 --voxels 60000 --steps 300 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 small scenes 300 steps: %.2f ms' % d['ms_per_step'])"
echo soak done
