cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3i
LGS_WIDE_TRACE=1 timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | grep "trace\|L0 3^3 512->512" | sort | uniq -c | sort -rn | head -12 > $O/wide_trace.txt; cat $O/wide_trace.txt
B="python bench.py --no-cpu-baseline --no-single-scene --no-secondary --no-roofline --steps 20 --warmup 6"
F=ffffffff
run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['stream_ms'])"; }
( run base A=1
  run "wgrad 64 CUs (low bits)" LGS_WGRAD_CUMASK=$F,$F
  run "wgrad 128 CUs (low bits)" LGS_WGRAD_CUMASK=$F,$F,$F,$F
  run "wgrad 64 / compute 192 disjoint" LGS_WGRAD_CUMASK=$F,$F LGS_COMPUTE_CUMASK=0,0,$F,$F,$F,$F,$F,$F
  run "wgrad 32 / compute 224 disjoint" LGS_WGRAD_CUMASK=$F LGS_COMPUTE_CUMASK=0,$F,$F,$F,$F,$F,$F,$F
  run "wgrad every 4th CU (64)" LGS_WGRAD_CUMASK=11111111,11111111,11111111,11111111,11111111,11111111,11111111,11111111
  run "wgrad every 4th / compute the rest" LGS_WGRAD_CUMASK=11111111,11111111,11111111,11111111,11111111,11111111,11111111,11111111 LGS_COMPUTE_CUMASK=eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee,eeeeeeee
  run base2 A=1 ) > $O/cumask.txt 2>&1; cat $O/cumask.txt
O2=$GRAFT_REPO_ROOT/gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O2/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --roctx --steps 1 --warmup 1 > $O2/prof.log 2>&1
DB=$(find $O2/prof -name "*.db" | head -1); cp $DB $O2/x.db; rm -rf $O2/prof
