cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_quantize.py tests/test_gpu_cluster.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do HOSTTIME_SCENES=1 python tools/hosttime.py 2>&1 | grep "host enqueue" | tail -2; done
