cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 400 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('400 steps: %.2f ms (median %.2f, p90 %.2f, max %.2f), single scene %.2f' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p90'], d['step_ms']['max'], d['secondary']['single_scene_ms']))"
timeout 300 python bench.py --dtype fp32 --steps 60 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 60 steps: %.2f ms (median %.2f)' % (d['ms_per_step'], d['step_ms']['median']))"
timeout 300 python bench.py --workload clip --steps 40 --warmup 3 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clip 40 steps: %.2f ms' % d['ms_per_step'])"
timeout 300 python bench.py --scenes 1 --steps 300 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 scene 300 steps: %.2f ms' % d['ms_per_step'])"
timeout 300 python bench.py --scenes 2 --voxels 60000 --steps 300 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 small scenes 300 steps: %.2f ms' % d['ms_per_step'])"
echo soak done
