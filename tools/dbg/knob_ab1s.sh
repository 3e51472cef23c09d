#!/bin/bash
# like knob_ab.sh, for the one-scene-per-step (latency) regime
R=$GRAFT_REPO_ROOT; cd $R
run() {
  env $1 python bench.py --scenes 1 --no-secondary --no-cpu-baseline --no-single-scene --no-roofline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); p=b['phases']['stream_ms']; h=b['phases']['host_enqueue_ms']
print('%-40s %.3f ms  fwd %.2f bwd %.2f fin %.2f opt %.2f | host %.2f' % (sys.argv[1], b['ms_per_step'], p['forward'], p['backward'], p['finalize'], p['optimizer'], sum(h.values())))" "$1"
}
run "LGS_NONE=0"
for s in "$@"; do run "$s"; done
run "LGS_NONE=0"
