"""A/B of the classifier's tile (tuning knob HEAD_TILE): 1x1 96 -> 200 (+ bias) forward on the 8-scene level-0 map, bf16 and fp32"""
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
for cin, cout in ((96, 200), (96, 160), (512, 200)):
    for dt in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        f = torch.randn(n, cin, device=DEV).to(dt)
        w = torch.randn(1, cin, cout, device=DEV) * 0.05
        b = torch.randn(1, cout, device=DEV)
        ref = None
        for rep in range(2):
            for tile in (0, 1):
                with engine.tuning(HEAD_TILE=tile):
                    y = km.conv_forward(f, w, b, False)
                    if ref is None:
                        ref = y.clone()
                    err = float((y.float() - ref.float()).abs().max())
                    t = timeit(lambda: km.conv_forward(f, w, b, False))
                byts = n * (cin + cout) * f.element_size()
                print("%s 1x1 %d->%d HEAD_TILE=%d: fwd %.3f ms (%.2f TB/s of the once-through bytes)  max |diff to tile 0| %.3g" % (
                    str(dt).split(".")[1], cin, cout, tile, t, byts / t / 1e9, err), flush=True)
