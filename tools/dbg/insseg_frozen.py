import sys, torch
sys.argv = ["bench.py"]
import bench
from languagegroundedsemseg_amd.synthetic import make_batch
dev = torch.device("cuda:0")
c_np, f_np, l_np = make_batch(list(range(8)), voxel=0.02, n_target=150000)
coords, feats, labels = [torch.from_numpy(a).to(dev) for a in (c_np, f_np, l_np)]
class A: sync_bn = 1; allreduce = "ring"
ctx = bench.make_ctx("insseg_frozen", "InsSegRes16UNet34C", coords, dev)
model, ddp, opt = bench.make_trainer("InsSegRes16UNet34C", torch.bfloat16, dev, 1, A, ctx)
clog = bench.ConvLog(); clog.patch()
import ctypes
clog.roctx = ctypes.CDLL("librocprofiler-sdk-roctx.so"); clog.mode = "roctx"
for i in range(6):
    bench.train_step(model, ddp, opt, coords, feats, labels, torch.bfloat16, i, ctx=ctx)
torch.cuda.synchronize()
