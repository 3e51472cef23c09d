cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -3 $O/pytest_wide.log
for d in 0 1 12 13; do echo "== LGS_WIDE_DBG=$d"; LGS_WIDE_DBG=$d timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-120; done > $O/wide_knockout.txt 2>&1; cat $O/wide_knockout.txt
for g in 2 4; do echo "== LGS_WIDE_GC64=$g"; LGS_WIDE_GC64=$g timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-120; done > $O/wide_gc.txt 2>&1; cat $O/wide_gc.txt
bash tools/run_pmc_wide.sh r3e/pmc > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r3e/pmc > $O/pmc_wide.txt 2>&1; grep -A36 "k_conv_wide" $O/pmc_wide.txt | head -40
rm -rf gpurun_out/r3e/pmc/*/
