"""A/B of k_pointwise (tuning knob POINTWISE) on the 1x1 launches of the 8-scene level-0 map: forward and dgrad per shape"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import MinkowskiEngine as ME
from languagegroundedsemseg_amd import engine
from languagegroundedsemseg_amd.synthetic import make_batch

DEV = "cuda:0"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


coords, _, _ = make_batch(list(range(8)), n_target=150000, shift_seed=0)
x = ME.SparseTensor(torch.zeros(coords.shape[0], 3, device=DEV), torch.from_numpy(coords).to(DEV))
m = x.coordinate_manager
km = m.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 1)
n = m.size(x.coordinate_map_key)
DT = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "fp32" else torch.bfloat16
for cin, cout, bias in ((96, 200, True), (128, 96, False), (96, 96, False)):
    torch.manual_seed(0)
    f = torch.randn(n, cin, device=DEV).to(DT)
    g = torch.randn(n, cout, device=DEV).to(DT)
    w = torch.randn(1, cin, cout, device=DEV) * 0.05
    b = torch.randn(1, cout, device=DEV) if bias else None
    byts = n * (cin + cout) * f.element_size()
    for rep in range(2):
        for on in (0, 2):
            with engine.tuning(POINTWISE=on):
                tf = timeit(lambda: km.conv_forward(f, w, b, False))
                td = timeit(lambda: km.conv_dgrad(g, w, False))
            print("1x1 %3d->%3d POINTWISE=%d: fwd %.3f ms (%.2f TB/s)  dgrad %.3f ms (%.2f TB/s)" % (
                cin, cout, on, tf, byts / tf / 1e9, td, byts / td / 1e9), flush=True)
