cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for m in "" inline; do echo -n "LGS_DBG_WGRAD='$m': "; LGS_DBG_WGRAD=$m python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], d['phases']['stream_ms'], 'single %.2f' % d['single_scene']['ms_per_step'])"; done; done
for m in "" inline; do echo -n "clip LGS_DBG_WGRAD='$m': "; LGS_DBG_WGRAD=$m python bench.py --workload clip --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-single-scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], d['phases']['stream_ms'])"; done
