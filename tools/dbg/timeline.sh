#!/bin/bash
# usage (GPU box): bash tools/dbg/timeline.sh <tag> [bench args]  -> gpurun_out/<tag>_timeline.txt (queue busy / idle-gap view of the step)
TAG=${1:-tl}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof_$TAG -o x -- python $R/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --steps 20 --warmup 6 "$@" > $O/prof_$TAG.log 2>&1
DB=$(find $O/prof_$TAG -name "*.db" | head -1)
python $R/tools/prof_timeline.py $DB > $O/${TAG}_timeline.txt 2>&1
rm -rf $O/prof_$TAG
cat $O/${TAG}_timeline.txt | tail -32
