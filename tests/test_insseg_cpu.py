"""downstream/insseg head against fixtures generated from the REFERENCE's insseg model
(insseg_models/insseg_res16unet.py imported through the MinkowskiEngine alias, tests/golden/make_fixtures.py insseg):
state-dict manifest, (offsets, logits) of the forward on the oracle backend, and the offset losses of
lib/pl_Trainer.py:271-299."""
import json
import os

import numpy as np
import torch

import MinkowskiEngine as ME
from helpers import Cfg, deterministic_init
from languagegroundedsemseg_amd.losses import instance_offset_losses
from languagegroundedsemseg_amd.models import load_model
from oracle.backend import OracleBackend

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_insseg_state_dict_matches_reference_manifest():
    man = json.load(open(os.path.join(G, "insseg_manifest.json")))
    for ref_name, ours in (("Res16UNet14A", "InsSegRes16UNet14A"), ("Res16UNet34C", "InsSegRes16UNet34C")):
        m = load_model(ours)(3, 20, Cfg())
        sd = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert sorted(sd) == sorted(man[ref_name]["state_dict"]), ref_name
        assert sum(p.numel() for p in m.parameters()) == man[ref_name]["num_parameters"]


def test_insseg_forward_and_losses_match_reference_fixture():
    fx = np.load(os.path.join(G, "insseg_res16unet14a_forward.npz"))
    prev = ME.set_backend(OracleBackend("c"))
    try:
        m = deterministic_init(load_model("InsSegRes16UNet14A")(3, 20, Cfg()), 42).train()
        x = ME.SparseTensor(torch.from_numpy(fx["feats"]), torch.from_numpy(fx["coords"]))
        off, logits, feats = m(x)
    finally:
        ME.set_backend(prev)
    assert np.abs(off.F.detach().numpy() - fx["offsets"]).max() < 1e-4
    assert np.abs(logits.F.detach().numpy() - fx["logits"]).max() < 1e-4
    nl, dl = instance_offset_losses(off.F, torch.from_numpy(fx["coords"][:, 1:]), torch.from_numpy(fx["centers"]),
                                    torch.from_numpy(fx["inst"]), float(fx["voxel"]))
    assert abs(float(nl) - float(fx["norm_loss"])) < 1e-5 and abs(float(dl) - float(fx["dir_loss"])) < 1e-5
