cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "bn or norm or batch" > $O/pytest_bn.log 2>&1; tail -3 $O/pytest_bn.log
for f in 0 1; do echo "== LGS_BN_FUSED=$f"; LGS_BN_FUSED=$f timeout 600 python tools/microbench.py 8 2>&1 | grep "BN"; done > $O/bn_ab.txt 2>&1; cat $O/bn_ab.txt
for f in 0 1; do for mb in 1000000 64 24 8; do echo "== LGS_BN_FUSED=$f MAX_MB=$mb"; LGS_BN_FUSED=$f LGS_BN_FUSED_MAX_MB=$mb python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['stream_ms'], d['phases']['host_enqueue_ms']['forward'], d['phases']['host_enqueue_ms']['backward'], d['single_scene']['ms_per_step'])"; [ $f = 0 ] && break; done; done > $O/bench_ab.txt 2>&1; cat $O/bench_ab.txt
