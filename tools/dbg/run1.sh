cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_teacher_forced.py tests/test_gpu_parity_r3.py "tests/test_gpu_parity_r2.py::test_res16unet34d_clip_step_vs_oracle" tests/test_gpu_model.py tests/test_gpu_rccl.py tests/test_gpu_ddp.py -m gpu -x -q -s > gpurun_out/r3a/pytest.log 2>&1; tail -15 gpurun_out/r3a/pytest.log
timeout 900 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -3 gpurun_out/r3a/bench.err; cut -c1-400 gpurun_out/r3a/bench.json
bash tools/probe_rocpd_schema.sh
