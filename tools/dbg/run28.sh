cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for sb in 1 0; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --same-device --backend gloo --sync-bn $sb --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-single-scene 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-1500
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --same-device --backend gloo --allreduce rs_ag --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-single-scene --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
