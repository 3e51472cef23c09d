cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o x -- python $R/bench.py --scenes 1 --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --steps 20 --warmup 6 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/prof_timeline.py $DB > $O/queue_timeline_1scene.txt 2>&1
rm -rf $O/prof
head -30 $O/queue_timeline_1scene.txt
