cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "wgrad or large_map or mfma_paths or reproducible or teacher_forced_34c or bf16_layerwise" 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2; do
for cfg in "24 256" "24 512" "64 512" "1000 512" "0 256"; do set -- $cfg
echo -n "rep $rep MAX_MB=$1 BLOCKS=$2: "; LGS_BN_FUSED_MAX_MB=$1 LGS_BN_FUSED_BLOCKS=$2 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f  single %.2f' % (d['ms_per_step'], d['single_scene']['ms_per_step']))"; done; done
