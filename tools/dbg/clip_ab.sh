cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  LGS_WIDE_WGRAD_INLINE=$v python bench.py --workload clip --steps 5 --warmup 3 --no-cpu-baseline --no-single-scene --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inline=$v clip ms/step', d['ms_per_step'])"
done
python bench.py --no-cpu-baseline --no-single-scene --no-secondary --steps 20 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline ms/step', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_clipprof
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --workload clip --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --roctx --steps 3 --warmup 3 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 6 > $O/kernel_stats_clip.txt 2>&1
rm -rf $O/prof
head -40 $O/kernel_stats_clip.txt | cut -c1-200
