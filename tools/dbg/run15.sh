cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3p
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -2 $O/pytest_wide.log
for c in 32768 16384 65536 8192; do echo "== LGS_WW_RANGE=$c"; LGS_WW_RANGE=$c timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | grep "3^3" | cut -c100-160; done > $O/wide_range.txt 2>&1; cat $O/wide_range.txt
