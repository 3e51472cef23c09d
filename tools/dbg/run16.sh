cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-single-scene --no-secondary --steps 20 --warmup 6 > $O/bench_ce.json 2>$O/bench_ce.err; cut -c1-400 $O/bench_ce.json
python bench.py --workload clip --no-cpu-baseline --no-single-scene --no-secondary --steps 6 --warmup 3 > $O/bench_clip.json 2>$O/bench_clip.err; cut -c1-400 $O/bench_clip.json
cd /tmp && export TMPDIR=/tmp
for w in ce clip; do
timeout 900 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof_$w -o x -- python $R/bench.py --workload $w --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --roctx --steps 3 --warmup 3 > $O/prof_$w.log 2>&1
DB=$(find $O/prof_$w -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 6 > $O/kernel_stats_$w.txt 2>&1
rm -rf $O/prof_$w
done
head -60 $O/kernel_stats_clip.txt
