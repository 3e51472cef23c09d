cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -3 $O/pytest_wide.log
for d in 0; do echo "== LGS_WIDE_DBG=$d"; LGS_WIDE_DBG=$d timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-120; done > $O/wide_knockout.txt 2>&1; cat $O/wide_knockout.txt
true
true
rm -rf gpurun_out/r3f/pmc/*/
