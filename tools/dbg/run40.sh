cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "4 0" "8 0" "8 1" "4 1"; do set -- $cfg; echo -n "queues=$1 prefetch=$2: "; GPU_MAX_HW_QUEUES=$1 LGS_BENCH_PREFETCH=$2 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], d['phases']['stream_ms'], 'single %.2f' % d['single_scene']['ms_per_step'])"; done; done
