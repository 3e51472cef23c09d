"""per-op GPU time of losses.sample_categories_for_balancing at the bench's size (1.2 M points, 200 classes)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from languagegroundedsemseg_amd.losses import sample_categories_for_balancing, fused_cross_entropy
import MinkowskiEngine as ME  # noqa

dev = "cuda:0"
n, L = 1205389, 200
torch.manual_seed(0)
logits = torch.randn(n, L, device=dev).to(torch.bfloat16).requires_grad_(True)
lab = torch.randint(-1, L, (n,), device=dev)
foc = torch.zeros(L, 3, dtype=torch.bool)
foc[:66, 0], foc[66:134, 1], foc[134:, 2] = True, True, True
foc = foc.to(dev)
for ratios in ((-1.0, -1.0), (0.5, 0.5)):
    def step():
        rows = fused_cross_entropy(logits, lab, -1, reduction="none")
        loss, stats, items = sample_categories_for_balancing(rows, lab, foc, ratios[0], ratios[1], split="stats")
        loss.backward()
        logits.grad = None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        step()
    e.record(); torch.cuda.synchronize()
    print("ratios %s: %.3f ms per loss fwd+bwd" % (ratios, s.elapsed_time(e) / 10))
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
