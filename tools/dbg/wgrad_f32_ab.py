"""stand-alone fp32 weight gradients of the level-0 / level-1 shapes: k_wgrad_f32_lds (rows staged through LDS) vs k_wgrad_f32"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import MinkowskiEngine as ME
from languagegroundedsemseg_amd import engine
from languagegroundedsemseg_amd.synthetic import make_batch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from microbench import timeit
DEV = "cuda:0"
coords, feats, labels = make_batch(list(range(8)), n_target=150000, shift_seed=0)
c = torch.from_numpy(coords).to(DEV)
n = coords.shape[0]
x = ME.SparseTensor(torch.zeros(n, 3, device=DEV), c)
mgr, k0 = x.coordinate_manager, x.coordinate_map_key
km = mgr.kernel_map_handle(k0, k0, 3)
for cin, cout in ((96, 96), (128, 96), (32, 32), (96, 200), (3, 32)):
    f = torch.randn(n, cin, device=DEV)
    g = torch.randn(n, cout, device=DEV)
    t = {}
    for mode in (1, 0):
        with engine.tuning(WGRAD_F32_LDS=mode):
            t[mode] = timeit(lambda: km.conv_wgrad(f, g, False), 5, 2)
    print("fp32 wgrad K=27 %3d->%3d rows %d: staged %.3f ms, pairwise loads %.3f ms" % (cin, cout, n, t[1], t[0]))
