cd $GRAFT_REPO_ROOT
for o in 0 1 2 3; do
  echo "== MASK_ORDER=$o"; LGS_MASK_ORDER=$o python tools/microbench.py 8 2>&1 | grep -E "bfloat16 +(96|128|32)"
done
for o in 0 1 2 3; do
  LGS_MASK_ORDER=$o python bench.py --no-cpu-baseline --no-single-scene --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MASK_ORDER=$o headline ms/step', d['ms_per_step'])"
done
