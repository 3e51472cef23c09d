cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r3b/pytest.log 2>&1; tail -8 gpurun_out/r3b/pytest.log
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --roctx --steps 2 --warmup 1 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1); ls -la $DB; cp $DB $O/x.db; rm -rf $O/prof
