cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for p in 1 0; do echo "== LGS_PW=$p"; LGS_PW=$p timeout 600 python tools/microbench.py 8 2>&1 | grep "bfloat16" ; done > $O/pw_ab.txt 2>&1; cat $O/pw_ab.txt
B="python bench.py --no-cpu-baseline --no-single-scene --no-secondary --steps 20 --warmup 6"
for p in 1 0 1 0; do echo "== bench LGS_PW=$p"; LGS_PW=$p $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['phases']['stream_ms'])"; done > $O/bench_ab.txt 2>&1; cat $O/bench_ab.txt
