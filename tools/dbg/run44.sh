cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$q bench.py --gpus 2 --same-device --backend gloo --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q:', d['ms_per_step'], [r['ddp']['syncbn_collective_ms'] for r in d['per_rank']])"
done
