"""A/B of k_conv_wide's DMA placement (tuning knob WIDE_SCHED) on the level-0 3^3 512 -> 512 launch of the 8-scene batch:
stand-alone forward / dgrad time per schedule, results compared bit for bit with schedule 0."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import MinkowskiEngine as ME
from languagegroundedsemseg_amd import engine
from languagegroundedsemseg_amd.synthetic import make_batch

DEV = "cuda:0"
scheds = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]


def timeit(fn, n=4, warm=2):
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
for cin, cout in ((512, 512), (544, 512)):
    km = m.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
    n = m.size(x.coordinate_map_key)
    M = km.export()[0].shape[0]
    torch.manual_seed(0)
    f = torch.randn(n, cin, device=DEV).bfloat16()
    g = torch.randn(n, cout, device=DEV).bfloat16()
    w = torch.randn(27, cin, cout, device=DEV) * 0.02
    ref = None
    for rep in range(2):
        for sc in scheds:
            with engine.tuning(WIDE_SCHED=sc):
                y = km.conv_forward(f, w, None, False)
                d = km.conv_dgrad(g, w, False)
                if ref is None:
                    ref = (y.clone(), d.clone())
                same = torch.equal(y, ref[0]) and torch.equal(d, ref[1])
                tf = timeit(lambda: km.conv_forward(f, w, None, False))
                td = timeit(lambda: km.conv_dgrad(g, w, False))
            flop = 2.0 * M * cin * cout
            print("%d->%d sched %d: fwd %.3f ms (%.0f TF)  dgrad %.3f ms (%.0f TF)  bit-identical to sched %d: %s" % (
                cin, cout, sc, tf, flop / tf / 1e9, td, flop / td / 1e9, scheds[0], same), flush=True)
