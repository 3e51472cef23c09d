"""stand-alone fp32 weight gradients of the level-0 shapes: k_wgrad_f32s_lds (staged rows, bf16-split products, WGRAD_F32_LDS=2),
k_wgrad_f32_lds (staged rows, exact fp32 MFMA, =1), k_wgrad_f32 (pairwise 4-byte loads, =0); relative L2 against the =0 result"""
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
for cin, cout in ((96, 96), (128, 96), (32, 32), (96, 200), (64, 64)):
    f = torch.randn(n, cin, device=DEV)
    g = torch.randn(n, cout, device=DEV)
    t, w = {}, {}
    for mode in (2, 1, 0):
        with engine.tuning(WGRAD_F32_LDS=mode):
            w[mode] = km.conv_wgrad(f, g, False).double()
            t[mode] = timeit(lambda: km.conv_wgrad(f, g, False), 5, 2)
    rel = float((w[2] - w[0]).norm() / w[0].norm())
    print("fp32 wgrad K=27 %3d->%3d rows %d: split %.3f ms, staged exact %.3f ms, pairwise loads %.3f ms; split vs exact rel-L2 %.2e, staged == pairwise: %s"
          % (cin, cout, n, t[2], t[1], t[0], rel, bool(torch.equal(w[1], w[0]))))
