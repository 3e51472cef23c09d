cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3x
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "0 0 512" "1 4 512" "1 8 512" "1 12 512" "1 24 512" "1 8 256" "1 12 256" "1 24 256" "1 24 128" "1 1000 128"; do set -- $cfg
echo -n "rep $rep FUSED=$1 MAX_MB=$2 BLOCKS=$3: "; LGS_BN_FUSED=$1 LGS_BN_FUSED_MAX_MB=$2 LGS_BN_FUSED_BLOCKS=$3 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f  single %.2f' % (d['ms_per_step'], d['single_scene']['ms_per_step']))"; done; done > $O/bn_sweep.txt 2>&1; cat $O/bn_sweep.txt
