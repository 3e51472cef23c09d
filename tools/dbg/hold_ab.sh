cd $GRAFT_REPO_ROOT
for cfg in "LGS_WGRAD_HOLD_ROWS=0" "LGS_WGRAD_HOLD_ROWS=600000" "LGS_WGRAD_HOLD_ROWS=600000 LGS_WGRAD_RELEASE_ROWS=400000" "LGS_WGRAD_HOLD_ROWS=600000 LGS_WGRAD_RELEASE_ROWS=30000" "LGS_WGRAD_HOLD_ROWS=200000 LGS_WGRAD_RELEASE_ROWS=150000" "LGS_WGRAD_HOLD_ROWS=0"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases']['stream_ms']; print('%-70s %.3f ms  fwd %.2f bwd %.2f fin %.2f opt %.2f' % ('$cfg', d['ms_per_step'], p['forward'], p['backward'], p['finalize'], p['optimizer']))"
done
timeout 600 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r3.py -q -k "block or bitwise or reproducible" 2>&1 | tail -3
