#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/...
# Everything profiles/rNN_* is made from: tests, smoke, the two bench lines, rocprofv3 kernel traces, PMC passes, microbench.
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
python bench.py --workload clip --steps 6 --warmup 3 > $O/bench_clip.json 2> $O/bench_clip.err; cut -c1-300 $O/bench_clip.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline --no-roofline --no-single-scene --steps 20 --warmup 6 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/prof_timeline.py $DB > $O/queue_timeline.txt 2>&1
python $R/tools/prof_summary.py $DB 26 > $O/kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof_clip -o x -- python $R/bench.py --workload clip --no-cpu-baseline --no-roofline --no-single-scene --steps 6 --warmup 3 > $O/prof_clip.log 2>&1
DBC=$(find $O/prof_clip -name "*.db" | head -1)
python $R/tools/prof_summary.py $DBC 9 > $O/kernel_stats_clip.txt 2>&1
rm -rf $O/prof $O/prof_clip
cd $R
bash tools/run_pmc.sh $TAG/pmc sq1 sq2 tcc1 fetch write > $O/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/$TAG/pmc > $O/pmc.txt 2>&1
python tools/pmc_traffic.py gpurun_out/$TAG/pmc $O/pmc_traffic.json > /dev/null 2>&1
( python tools/microbench.py 8; python tools/microbench.py wgrad; python tools/microbench.py clip; python tools/microbench.py wide ) > $O/microbench.txt 2>&1
head -12 $O/queue_timeline.txt
