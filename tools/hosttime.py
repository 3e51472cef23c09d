"""How long does the HOST need to enqueue one training step (no GPU sync inside)?"""
import sys, time
import torch
sys.argv = ["bench.py"]
import bench
import MinkowskiEngine as ME
from languagegroundedsemseg_amd.ddp import BucketedDDP
from languagegroundedsemseg_amd.synthetic import make_batch

dev = torch.device("cuda:0")
import os
B = int(os.environ.get("HOSTTIME_SCENES", "8"))
coords_np, feats_np, labels_np = make_batch(list(range(B)), voxel=0.02, n_target=150000)
coords, feats, labels = [torch.from_numpy(a).to(dev) for a in (coords_np, feats_np, labels_np)]
model = bench.build(dev, torch.bfloat16)
ddp = BucketedDDP(model)
from languagegroundedsemseg_amd.ddp import FlatSGD
opt = FlatSGD(ddp, lr=1e-2, momentum=0.9, dampening=0.1, weight_decay=1e-4)
for i in range(3):
    bench.train_step(model, ddp, opt, coords, feats, labels, torch.bfloat16, i)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    bench.train_step(model, ddp, opt, coords, feats, labels, torch.bfloat16, 10 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.1f ms, then GPU drain %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):   # backward on this thread, so that the profiler sees it
    bench.train_step(model, ddp, opt, coords, feats, labels, torch.bfloat16, 19)
    torch.cuda.synchronize()
    pr.enable()
    for i in range(5):
        bench.train_step(model, ddp, opt, coords, feats, labels, torch.bfloat16, 20 + i)
    pr.disable()
torch.cuda.synchronize()
print("== 5 steps, by cumulative time")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
print("== 5 steps, by own time")
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
