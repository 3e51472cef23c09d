import torch, sys
sys.path.insert(0, "tests")
import MinkowskiEngine as ME
from helpers import Cfg, deterministic_init
from languagegroundedsemseg_amd import models
from languagegroundedsemseg_amd.models import load_model
from languagegroundedsemseg_amd.synthetic import make_batch
from torch.nn.modules import module as _m
m = load_model("Res16UNet34C")(3, 20, Cfg()).to("cuda:0").train()
blk = m.block1[0]
be = models.ME.get_backend()
print("backend", type(be), getattr(be, "conv_bn_stats", None), hasattr(be, "bn_forward"), hasattr(be, "side_stream"))
print("global hooks", len(_m._global_forward_hooks), len(_m._global_forward_pre_hooks), len(_m._global_backward_hooks))
print("types", type(blk.conv1) is models.ME.MinkowskiConvolution, type(blk.norm1) is models.ME.MinkowskiBatchNorm, models.ME.MinkowskiConvolution, type(blk.conv1))
print("plain", models._plain(blk), models._plain(blk.conv1), models._plain(blk.norm1), models._plain(blk.norm1.bn))
print("fused flag", models._BLOCK_FUSED, blk.cat_up, torch.is_grad_enabled())
coords, feats, labels = make_batch([5], voxel=0.05, n_target=9000)
x = ME.SparseTensor(torch.from_numpy(feats).to("cuda:0").bfloat16(), torch.from_numpy(coords).to("cuda:0"))
y = m.bn0(m.conv0p1s1(x), relu=True)
print("ok?", models._block_fast_path_ok(blk, y))
