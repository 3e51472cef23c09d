cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -3 $O/pytest_wide.log
for p in 0 1; do echo "== LGS_WIDE_PIPE=$p"; LGS_WIDE_PIPE=$p timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-150; done > $O/wide_pipe.txt 2>&1; cat $O/wide_pipe.txt
LGS_WIDE_TRACE=1 timeout 600 python tools/microbench.py wide 2>&1 | grep "trace" | sort | uniq -c | sort -rn | head -4
