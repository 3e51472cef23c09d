cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for q in "" 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; if [ -z "$q" ]; then python tools/dbg/hostblock.py 2>&1 | grep "step [345] "; else GPU_MAX_HW_QUEUES=$q python tools/dbg/hostblock.py 2>&1 | grep "step [345] "; fi; done
for q in 4 8; do echo -n "bench GPU_MAX_HW_QUEUES=$q: "; GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], d['phases']['stream_ms'], 'single %.2f' % d['single_scene']['ms_per_step'])"; done
