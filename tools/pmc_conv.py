"""Tiny driver for PMC passes: L0 96->96 3^3 forward / dgrad / wgrad (bf16) on an 8-scene batch, 3 launches each."""
import torch
import MinkowskiEngine as ME
from languagegroundedsemseg_amd.synthetic import make_batch

DEV = "cuda:0"
coords, feats, labels = make_batch(list(range(8)), n_target=150000, shift_seed=0)
c = torch.from_numpy(coords).to(DEV)
n = coords.shape[0]
x = ME.SparseTensor(torch.zeros(n, 3, device=DEV), c)
km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
f = torch.randn(n, 96, device=DEV).bfloat16()
g = torch.randn(n, 96, device=DEV).bfloat16()
w = torch.randn(27, 96, 96, device=DEV) * 0.05
for _ in range(3):
    km.conv_forward(f, w, None, False)
    km.conv_wgrad(f, g, False)
torch.cuda.synchronize()
print("done", n)
