cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for m in 0 big 1; do echo -n "LGS_CONV_BN_STATS=$m: "; LGS_CONV_BN_STATS=$m python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-single-scene --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], d['phases']['stream_ms'])"; done; done
