"""time of the fused cross-entropy on the head's logits (1.2 M x 200): loss pass and gradient pass, bf16 and fp32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import MinkowskiEngine as ME
be = ME.get_backend()
n = 1205389
for c in (200, 20):
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(n, c, device="cuda").to(dt)
        lab = torch.randint(-1, c, (n,), device="cuda")
        g = torch.ones((), device="cuda")
        def t(fn, k=20):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(k): fn()
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / k
        _, _, inv = be.cross_entropy(x, lab, -1, want_grad=False)
        tf = t(lambda: be.cross_entropy(x, lab, -1, want_grad=False, inv_valid=inv))
        tb = t(lambda: be.cross_entropy(x, lab, -1, grad_scale=g, inv_valid=inv))
        b = n * c * x.element_size()
        print("%s C=%d: loss pass %.3f ms (%.2f TB/s)  gradient pass %.3f ms (%.2f TB/s)" % (str(dt).split(".")[1], c, tf, b / tf / 1e9, tb, 2 * b / tb / 1e9))
