cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
HOSTTIME_SCENES=1 python tools/hosttime.py > $O/hosttime1.txt 2>&1
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 6 > $O/bench_ce.json 2>$O/bench_ce.err; cut -c1-300 $O/bench_ce.json; grep -o '"single_scene.*' $O/bench_ce.json | cut -c1-300
