cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "large_map or wide or big" > $O/pytest_wide.log 2>&1; tail -5 $O/pytest_wide.log
timeout 600 python tools/microbench.py wide > $O/wide_new.txt 2>&1; cat $O/wide_new.txt
LGS_WIDE_CFG=16 timeout 600 python tools/microbench.py wide > $O/wide_old.txt 2>&1; cat $O/wide_old.txt
timeout 1500 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -x -k 34d > $O/pytest_tf.log 2>&1; tail -5 $O/pytest_tf.log
