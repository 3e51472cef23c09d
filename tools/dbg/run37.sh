cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/profg -o x -- python $R/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --steps 20 --warmup 6 > $O/profg.log 2>&1
DB=$(find $O/profg -name "*.db" | head -1)
cd $R/tools && python prof_overlap.py $DB > $O/overlap.txt 2>&1
rm -rf $O/profg
cat $O/overlap.txt
