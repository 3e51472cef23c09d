cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/profi -o x -- python $R/tools/dbg/insseg_frozen.py > $O/profi.log 2>&1
DB=$(find $O/profi -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 6 > $O/kernel_stats_insseg_frozen.txt 2>&1
rm -rf $O/profi
tail -5 $O/profi.log; head -24 $O/kernel_stats_insseg_frozen.txt | cut -c1-200
