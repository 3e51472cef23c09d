cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3m
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -4 $O/pytest_wide.log
for c in 1 0; do echo "== LGS_WGRAD_WIDE=$c"; LGS_WGRAD_WIDE=$c timeout 600 python tools/microbench.py wide 2>&1 | grep -v amdgpu.ids | grep "3^3" | cut -c1-160; done > $O/wide_ab.txt 2>&1; cat $O/wide_ab.txt
timeout 1500 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -x -k 34d > $O/pytest_tf.log 2>&1; tail -3 $O/pytest_tf.log
