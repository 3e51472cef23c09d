import sys, time, torch
sys.argv = ["bench.py"]
import bench
import MinkowskiEngine as ME
from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
from languagegroundedsemseg_amd.losses import fused_cross_entropy
from languagegroundedsemseg_amd.synthetic import make_batch
dev = torch.device("cuda:0")
c_np, f_np, l_np = make_batch(list(range(8)), voxel=0.02, n_target=150000)
coords, feats, labels = [torch.from_numpy(a).to(dev) for a in (c_np, f_np, l_np)]
model = bench.build(dev, torch.bfloat16)
ddp = BucketedDDP(model); opt = FlatSGD(ddp, lr=1e-2, momentum=0.9, dampening=0.1, weight_decay=1e-4)
for i in range(5):
    bench.train_step(model, ddp, opt, coords, feats, labels, torch.bfloat16, i)
torch.cuda.synchronize()
E0 = torch.cuda.Event(enable_timing=True); E0.record(); torch.cuda.synchronize(); T0 = time.perf_counter()
log = []
main = torch.cuda.current_stream()
ds = torch.cuda.Stream()
for i in range(8):
    h = [time.perf_counter()]
    with torch.cuda.stream(ds):
        c = coords.clone(); c[:, 1] += 3; f = feats.to(torch.bfloat16)
        h.append(time.perf_counter())
        sinput = ME.SparseTensor(f, c)
    h.append(time.perf_counter())
    main.wait_stream(ds)
    ddp.zero_grad()
    logits, _ = model(sinput)
    loss = fused_cross_entropy(logits.F, labels, ignore_index=-1)
    h.append(time.perf_counter())
    loss.backward()
    h.append(time.perf_counter())
    ddp.finalize(); opt.step()
    h.append(time.perf_counter())
    e = torch.cuda.Event(enable_timing=True); e.record()
    log.append((h, e))
torch.cuda.synchronize()
for i, (h, e) in enumerate(log):
    print("step %d host(ms since T0): start %.2f  data %.2f  insert-done %.2f  fwd-enq %.2f  bwd-enq %.2f  opt-enq %.2f | GPU end of step %.2f" % (
        i, *[(x - T0) * 1e3 for x in h], E0.elapsed_time(e)))
