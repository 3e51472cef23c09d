cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "insseg or odd or ragged or single_voxel or conv_fwd_bwd or mfma_paths" 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --no-cpu-baseline --no-roofline --no-single-scene --steps 5 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); i=d['insseg']; print('insseg full %.2f' % i['full']['ms_per_step'], i['full']['phases']['stream_ms']); print('frozen %.2f' % i['frozen_trunk']['ms_per_step'], i['frozen_trunk']['phases']['stream_ms'])"
