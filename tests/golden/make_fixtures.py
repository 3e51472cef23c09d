"""Generates the committed golden fixtures by IMPORTING THE REFERENCE'S PYTHON (this container only:
/root/reference does not exist on the GPU box).  Fixtures are data only -- inputs and expected outputs.

    python tests/golden/make_fixtures.py

1. models_manifest.json   state-dict keys/shapes + parameter counts of the reference's model classes
                          (models/res16unet.py, models/clip_models.py) instantiated through the
                          MinkowskiEngine alias package.
2. contrastive_loss.npz   inputs/outputs of the reference's own ContrastiveLanguageLoss.feat_dist +
                          hinge (lib/losses/ContrastiveLanguageLoss.py:73-95,185-192) on CPU, with the
                          sampled negative indices made explicit (the reference's sampling is thread-racy).
3. res16unet14a_forward.npz / res16unet34c_forward.npz
                          logits + features of the REFERENCE's forward code (res16unet.py:196-270,
                          resnet_block.py:41-57) run on the CPU oracle backend with name-keyed
                          deterministic weights; pins the build's models.py dataflow.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import MinkowskiEngine as ME  # noqa: E402
from helpers import Cfg, deterministic_init, small_scene  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402


def manifest():
    from models import load_model
    out = {}
    for name in ["Res16UNet14A", "Res16UNet18A", "Res16UNet34C", "Res16UNet34CR", "Res16UNet34CR_Proj", "Res16UNet34D"]:
        m = load_model(name)(3, 200, Cfg())
        sd = m.state_dict()
        out[name] = {"num_parameters": int(sum(p.numel() for p in m.parameters())),
                     "state_dict": [[k, list(v.shape)] for k, v in sd.items()]}
    with open(os.path.join(HERE, "models_manifest.json"), "w") as f:
        json.dump(out, f)
    print("manifest:", {k: v["num_parameters"] for k, v in out.items()})


def contrastive():
    # torchmetrics / joblib-free import of the loss module
    for mod in ("torchmetrics",):
        if mod not in sys.modules:
            sys.modules[mod] = types.SimpleNamespace(Metric=object)
    from lib.losses.ContrastiveLanguageLoss import ContrastiveLanguageLoss
    cfg = types.SimpleNamespace(ignore_label=-1, num_negative_samples=3, contrast_neg_thresh=0.6, contrast_pos_thresh=0.0,
                                contrast_neg_weight=1.0, instance_augmentation_color_aug_prob=0.0, scannet_path="/nonexistent",
                                projection_model_path="none", representation_distance_type="cos", clip_uniform_sampling=True)
    out = {}
    for tag, C in (("c512", 512), ("c96", 96)):
        loss = ContrastiveLanguageLoss(cfg, 200, feature_dim=C)
        g = torch.Generator().manual_seed(7 + C)
        N, K = 257, 3
        F = torch.randn(N, C, generator=g)
        T = torch.randn(200, C, generator=g)
        labels = torch.randint(0, 200, (N,), generator=g)
        labels[torch.rand(N, generator=g) < 0.1] = -1
        neg = torch.randint(0, 199, (N, K), generator=g)
        lab_safe = labels.clamp_min(0)
        neg = neg + (neg >= lab_safe[:, None]).long()          # uniform over the other 199 classes
        pos_samples = T[lab_safe].view(N, 1, C)
        neg_samples = T[neg.view(-1)].view(N, K, C)
        d_pos = loss.feat_dist(F, pos_samples, labels)
        d_neg = loss.feat_dist(F, neg_samples, labels)
        pos_loss = torch.relu(d_pos - cfg.contrast_pos_thresh)
        neg_loss = torch.relu(cfg.contrast_neg_thresh - d_neg)
        total = pos_loss.mean() + neg_loss.mean() * cfg.contrast_neg_weight
        for k, v in dict(F=F, T=T, labels=labels, neg=neg, d_pos=d_pos, d_neg=d_neg, pos_loss=pos_loss, neg_loss=neg_loss,
                         total=total.reshape(1)).items():
            out["%s_%s" % (tag, k)] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "contrastive_loss.npz"), **out)
    print("contrastive fixture written")


def forward_fixture(name, seed, n):
    from models import load_model
    prev = ME.set_backend(OracleBackend("c"))
    try:
        torch.manual_seed(0)
        m = deterministic_init(load_model(name)(3, 20, Cfg()), 42)
        m.train()
        coords = small_scene(seed, n=n)
        rng = np.random.default_rng(seed)
        feats = rng.uniform(-0.5, 0.5, (coords.shape[0], 3)).astype(np.float32)
        x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords))
        logits, fmap = m(x)
        np.savez_compressed(os.path.join(HERE, "%s_forward.npz" % name.lower()), coords=coords, feats=feats,
                            logits=logits.F.detach().numpy(), fmap=fmap.F.detach().numpy().astype(np.float16),
                            running_mean_bn0=m.bn0.bn.running_mean.numpy(),
                            running_mean_b1n1=m.block1[0].norm1.bn.running_mean.numpy())
        print(name, "forward fixture:", coords.shape, logits.F.shape, float(logits.F.abs().mean()))
    finally:
        ME.set_backend(prev)


if __name__ == "__main__":
    manifest()
    contrastive()
    forward_fixture("Res16UNet14A", 3, 1500)
    forward_fixture("Res16UNet34C", 5, 1200)
