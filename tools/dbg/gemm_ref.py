"""library GEMM (rocBLAS / hipBLASLt through torch.mm) on the 1x1 shapes of level 0, for comparison with the engine's kernels"""
import torch
n = 1205389
def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
for dt in (torch.bfloat16, torch.float32):
    for cin, cout in ((96, 200), (200, 96), (128, 96), (96, 128)):
        x = torch.randn(n, cin, device="cuda", dtype=dt)
        w = torch.randn(cin, cout, device="cuda", dtype=dt)
        b = torch.randn(cout, device="cuda", dtype=dt)
        out = torch.empty(n, cout, device="cuda", dtype=dt)
        tm = t(lambda: torch.mm(x, w, out=out))
        ta = t(lambda: torch.addmm(b, x, w, out=out))
        byts = n * (cin + cout) * x.element_size()
        print("%s [%d x %d] x [%d x %d]: mm %.3f ms (%.2f TB/s)  addmm %.3f ms" % (str(dt).split(".")[1], n, cin, cin, cout, tm, byts / tm / 1e9, ta))
