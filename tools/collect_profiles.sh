#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag> [skip-tests]   -> gpurun_out/<tag>/...
# Everything profiles/rNN_* is made from: tests, smoke, the two bench lines, rocprofv3 kernel traces (shape-keyed through the
# ROCTx ranges of bench.py --roctx), the queue timeline, PMC passes (own runs, no other trace domain), micro-benchmarks.
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export PYTHONPATH=$R
if [ -z "$2" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
LGS_BENCH_DETAILS=$O/bench_full.json python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; wc -c $O/bench.json
LGS_BENCH_DETAILS=$O/bench_clip_full.json python bench.py --workload clip --steps 6 --warmup 3 --no-secondary > $O/bench_clip.json 2> $O/bench_clip.err; cut -c1-300 $O/bench_clip.json
HOSTTIME_SCENES=1 python tools/hosttime.py > $O/hosttime_1scene.txt 2>&1
HOSTTIME_SCENES=8 python tools/hosttime.py > $O/hosttime_8scenes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export LGS_BENCH_DETAILS=$O/profiled_run_details.json   # (the profiled runs below: their details are not results)
# queue timeline: plain kernel trace of the default workload
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --settle 0 --steps 20 --warmup 6 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/prof_timeline.py $DB > $O/queue_timeline.txt 2>&1
rm -rf $O/prof
# per-(kernel, grid, shape) statistics: kernel trace + ROCTx ranges around every conv / dgrad / wgrad engine call
for w in ce clip; do
  timeout 900 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof_$w -o x -- python $R/bench.py --workload $w --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --settle 0 --roctx --steps 3 --warmup 3 > $O/prof_$w.log 2>&1
  DB=$(find $O/prof_$w -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB 6 > $O/kernel_stats_$w.txt 2>&1
  rm -rf $O/prof_$w
done
# the fp32 parity path, per launch shape
timeout 900 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof_f32 -o x -- python $R/bench.py --dtype fp32 --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --settle 0 --roctx --steps 3 --warmup 3 > $O/prof_f32.log 2>&1
DB=$(find $O/prof_f32 -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 6 > $O/kernel_stats_fp32.txt 2>&1
rm -rf $O/prof_f32
# the same step with every kernel alone on the machine (weight gradients on the compute stream): stand-alone durations per shape
timeout 900 env LGS_DBG_WGRAD=inline rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof_sa -o x -- python $R/bench.py --workload ce --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --settle 0 --roctx --steps 3 --warmup 3 > $O/prof_sa.log 2>&1
DB=$(find $O/prof_sa -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 6 > $O/kernel_stats_standalone.txt 2>&1
rm -rf $O/prof_sa
python $R/tools/instep_table.py $O/kernel_stats_ce.txt $O/kernel_stats_standalone.txt > $O/instep_table.txt 2>&1
cd $R
bash tools/run_pmc.sh $TAG/pmc sq1 sq2 tcc1 fetch write > $O/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/$TAG/pmc > $O/pmc.txt 2>&1
python tools/pmc_traffic.py gpurun_out/$TAG/pmc $O/pmc_traffic.json > /dev/null 2>&1
rm -rf gpurun_out/$TAG/pmc/*/
bash tools/run_pmc_wide.sh $TAG/pmcw > $O/pmc_wide.log 2>&1
python tools/pmc_summary.py gpurun_out/$TAG/pmcw > $O/pmc_wide.txt 2>&1
python tools/pmc_traffic.py gpurun_out/$TAG/pmcw $O/pmc_traffic_wide.json k_conv_wide > /dev/null 2>&1
rm -rf gpurun_out/$TAG/pmcw/*/
( python tools/microbench.py 8; python tools/microbench.py wgrad; python tools/microbench.py clip; python tools/microbench.py wide ) 2>&1 | grep -v amdgpu.ids > $O/microbench.txt
head -12 $O/queue_timeline.txt
