cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3g
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3g/bench.json'))
print("main", d['ms_per_step'], d['roofline']['frac'], "single", d['single_scene']['ms_per_step'])
print("fp32", d['fp32']['ms_per_step'], "clip", d['clip']['ms_per_step'], d['clip']['phases']['stream_ms'], "insseg", d['insseg']['full']['ms_per_step'], d['insseg']['frozen_trunk']['ms_per_step'])
for t in d['clip']['roofline']['discovery_step']['top_shapes']: print(t)
print(d['clip']['roofline']['avg_launch_ms'], d['clip']['roofline']['mfma_tflops_on_real_pairs'])
PY
O2=$GRAFT_REPO_ROOT/gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O2/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --roctx --steps 4 --warmup 2 > $O2/prof.log 2>&1
DB=$(find $O2/prof -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 6 > $O2/kernel_stats.txt 2>&1; head -30 $O2/kernel_stats.txt; rm -rf $O2/prof
