"""A/B of the k_conv_gather tile for 128 output channels on the big maps (knob BIG4_CFG): stand-alone forward / dgrad of the level-0
shapes with 128 output channels (dgrad of block8.0.conv1: 96 -> 128; forward 128 -> 128 at level 1)"""
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
for cfg in (7, 3, 7, 3):
    with engine.tuning(BIG4_CFG=cfg):
        for cin, cout in ((96, 128), (128, 128), (64, 128)):
            f = torch.randn(n, cin, device=DEV).bfloat16()
            g = torch.randn(n, cout, device=DEV).bfloat16()
            w = torch.randn(27, cin, cout, device=DEV) * 0.05
            wt = torch.randn(27, cout, cin, device=DEV) * 0.05
            tf = timeit(lambda: km.conv_forward(f, w, None, False), 10, 3)
            td = timeit(lambda: km.conv_dgrad(f, wt, False), 10, 3)      # dgrad producing 128 channels from `cin`-wide gradients
            print("BIG4_CFG=%d  %3d->%3d forward %.3f ms   dgrad (gout %d -> gin %d) %.3f ms" % (cfg, cin, cout, tf, cin, cout, td))
