cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3h
for c in 17 16; do echo "== LGS_WIDE_CFG=$c"; LGS_WIDE_CFG=$c timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | cut -c1-150; done > $O/wide_ab.txt 2>&1; cat $O/wide_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "picklable or insseg" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
