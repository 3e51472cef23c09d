cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for c in 32 28 24 20 16; do echo -n "LGS_PS_CUS=$c: "; LGS_PS_CUS=$c python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-single-scene --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'], d['phases']['stream_ms'])"; done; done
