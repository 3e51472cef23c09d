#!/bin/bash
# usage (GPU box): bash tools/dbg/knob_ab.sh "LGS_A=1" "LGS_B=2 LGS_C=3" ...   -> one bench line (ms/step, phases) per setting, default first and last
R=$GRAFT_REPO_ROOT; cd $R
run() {
  env $1 python bench.py --no-secondary --no-cpu-baseline --no-single-scene --no-roofline --steps 30 --warmup 6 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); p=b['phases']['stream_ms']
print('%-40s %.3f ms  fwd %.2f bwd %.2f fin %.2f opt %.2f' % (sys.argv[1], b['ms_per_step'], p['forward'], p['backward'], p['finalize'], p['optimizer']))" "$1"
}
run "LGS_NONE=0"
for s in "$@"; do run "$s"; done
run "LGS_NONE=0"
