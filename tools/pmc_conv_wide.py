"""PMC driver for the wide-channel case (BASELINE configs[2]): L0 3^3 512->512 forward + wgrad (bf16), 2-scene batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import MinkowskiEngine as ME
from languagegroundedsemseg_amd.synthetic import make_batch

DEV = "cuda:0"
coords, feats, labels = make_batch(list(range(2)), n_target=150000, shift_seed=0)
c = torch.from_numpy(coords).to(DEV)
n = coords.shape[0]
x = ME.SparseTensor(torch.zeros(n, 3, device=DEV), c)
km = x.coordinate_manager.kernel_map_handle(x.coordinate_map_key, x.coordinate_map_key, 3)
f = torch.randn(n, 512, device=DEV).bfloat16()
g = torch.randn(n, 512, device=DEV).bfloat16()
w = torch.randn(27, 512, 512, device=DEV) * 0.02
for _ in range(2):
    km.conv_forward(f, w, None, False)
    km.conv_wgrad(f, g, False)
torch.cuda.synchronize()
print("done", n)
