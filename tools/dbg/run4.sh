cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -3 $O/pytest_wide.log
for d in 0 1 4 8 12 13; do echo "== LGS_WIDE_DBG=$d"; LGS_WIDE_DBG=$d timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-120; done > $O/wide_knockout.txt 2>&1; cat $O/wide_knockout.txt
bash tools/run_pmc_wide.sh r3d/pmc > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r3d/pmc > $O/pmc_wide.txt 2>&1; grep -A45 "k_conv_wide" $O/pmc_wide.txt | head -60
rm -rf gpurun_out/r3d/pmc/*/
