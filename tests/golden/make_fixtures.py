"""Generates the committed golden fixtures by IMPORTING THE REFERENCE'S PYTHON (this container only:
/root/reference does not exist on the GPU box).  Fixtures are data only -- inputs and expected outputs.

    python tests/golden/make_fixtures.py

1. models_manifest.json   state-dict keys/shapes + parameter counts of the reference's model classes
                          (models/res16unet.py, models/clip_models.py) instantiated through the
                          MinkowskiEngine alias package.
2. contrastive_loss.npz   inputs/outputs of the reference's own ContrastiveLanguageLoss.feat_dist +
                          hinge (lib/losses/ContrastiveLanguageLoss.py:73-95,185-192) on CPU, with the
                          sampled negative indices made explicit (the reference's sampling is thread-racy).
2d. contrastive_distances.npz  the same for representation_distance_type 'l1' / 'l2' (:79-86).
2b. feature_sim.npz       the reference's feature_sim (lib/losses/utils.py:80-103, cosine branch) + argmax on the
                          contrastive fixture's features / anchors (2-D anchors and the 3-D attribute layout).
2c. balancing.npz         the reference's sample_categories_for_balancing (lib/losses/utils.py:13-77) on fixed
                          labels: kept points per class (recovered with one-hot losses), the head/common/tail
                          split of a random loss and loss_items.
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


def distance_fixture():
    """2d. contrastive_distances.npz: the reference's feat_dist + hinge for representation_distance_type 'l1' and 'l2'
    (lib/losses/ContrastiveLanguageLoss.py:79-86) on the inputs of contrastive_loss.npz.  Note the reference's 'l1' is the
    SIGNED sum of differences (no abs) -- reproduced as written."""
    for mod in ("torchmetrics",):
        if mod not in sys.modules:
            sys.modules[mod] = types.SimpleNamespace(Metric=object)
    from lib.losses.ContrastiveLanguageLoss import ContrastiveLanguageLoss
    fx = np.load(os.path.join(HERE, "contrastive_loss.npz"))
    out = {}
    for dist in ("l1", "l2"):
        cfg = types.SimpleNamespace(ignore_label=-1, num_negative_samples=3, contrast_neg_thresh=0.6, contrast_pos_thresh=0.0,
                                    contrast_neg_weight=1.0, instance_augmentation_color_aug_prob=0.0, scannet_path="/nonexistent",
                                    projection_model_path="none", representation_distance_type=dist, clip_uniform_sampling=True)
        for tag, C in (("c512", 512), ("c96", 96)):
            loss = ContrastiveLanguageLoss(cfg, 200, feature_dim=C)
            F, T = torch.from_numpy(fx[tag + "_F"]), torch.from_numpy(fx[tag + "_T"])
            labels, neg = torch.from_numpy(fx[tag + "_labels"]), torch.from_numpy(fx[tag + "_neg"])
            N, K = neg.shape
            pos_samples = T[labels.clamp_min(0)].view(N, 1, C)
            neg_samples = T[neg.view(-1)].view(N, K, C)
            d_pos = loss.feat_dist(F, pos_samples, labels)
            d_neg = loss.feat_dist(F, neg_samples, labels)
            pos_loss = torch.relu(d_pos - cfg.contrast_pos_thresh)
            neg_loss = torch.relu(cfg.contrast_neg_thresh - d_neg)
            total = pos_loss.mean() + neg_loss.mean() * cfg.contrast_neg_weight
            for k, v in dict(d_pos=d_pos, d_neg=d_neg, pos_loss=pos_loss, neg_loss=neg_loss, total=total.reshape(1)).items():
                out["%s_%s_%s" % (dist, tag, k)] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "contrastive_distances.npz"), **out)
    print("distance fixture:", sorted(out)[:4], "...")


def ce_and_latent_fixture():
    """2e. contrastive_ce_latent.npz
    (a) the reference's ContrastiveLanguageCELoss.forward (lib/losses/ContrastiveLanguageLoss.py:196-237) on CPU, 'cos' and 'l2';
    (b) the reference's ContrastiveLanguageLoss.forward with (category, attribute) labels and instance_augmentation='latent'
        (:62-70,:149-181) run on CPU: its two `torch.cuda.FloatTensor(...)` allocations are pointed at the CPU allocator for the
        duration of the call, the joblib pool is one worker (sequential, in unique-target order), and the host RNG draws it makes
        -- per unique target "augment?" / "which attribute", and the negatives -- are RECORDED so that the engine-side loss can be
        given the same decisions explicitly (the reference's own draw order is not reproducible in a thread pool).
        Every row's original attribute slot is 0, so no relabelled row is picked up a second time by a later unique target
        (an order-dependent quirk of the reference loop that is not part of the fixture)."""
    for mod in ("torchmetrics",):
        if mod not in sys.modules:
            sys.modules[mod] = types.SimpleNamespace(Metric=object)
    import random
    import lib.losses.ContrastiveLanguageLoss as RL
    import models.projection_models as RP
    out = {}
    base = dict(ignore_label=-1, num_negative_samples=3, contrast_neg_thresh=0.6, contrast_pos_thresh=0.0, contrast_neg_weight=1.0,
                instance_augmentation_color_aug_prob=0.0, scannet_path="/nonexistent", projection_model_path="none",
                clip_uniform_sampling=True)
    fx = np.load(os.path.join(HERE, "contrastive_loss.npz"))
    # ---- (a)
    for dist in ("cos", "l2"):
        cfg = types.SimpleNamespace(representation_distance_type=dist, **base)
        for tag in ("c512", "c96"):
            crit = RL.ContrastiveLanguageCELoss(cfg, 200)
            F = torch.from_numpy(fx[tag + "_F"]).clone().requires_grad_(True)
            T, labels = torch.from_numpy(fx[tag + "_T"]), torch.from_numpy(fx[tag + "_labels"])
            loss, z, loss2 = crit(F, labels, T)
            loss.backward()
            out["ce_%s_%s_loss" % (dist, tag)] = loss.detach().reshape(1).numpy()
            out["ce_%s_%s_gradF" % (dist, tag)] = F.grad.numpy().astype(np.float32)
    # ---- (b)
    C, N, K, L, A = 96, 400, 3, 200, 9
    g = torch.Generator().manual_seed(23)
    F0 = torch.randn(N, C, generator=g)
    T3 = torch.randn(L, A, C, generator=g)
    cat = torch.randint(0, 12, (N,), generator=g)                      # a dozen categories, ~33 rows each
    cat[torch.rand(N, generator=g) < 0.1] = -1
    labels0 = torch.stack([cat, torch.zeros_like(cat)], 1)
    cfg = types.SimpleNamespace(representation_distance_type="cos", instance_augmentation="latent", **base)
    cfg.instance_augmentation_color_aug_prob = 0.7
    torch.manual_seed(5)
    crit = RL.ContrastiveLanguageLoss(cfg, L, feature_dim=C)
    crit.num_cores = 1
    crit.augment_categories = torch.tensor([0, 2, 3, 5, 7, 8, 11])
    plan_on, plan_attr = np.zeros(L * A, bool), np.zeros(L * A, np.int64)
    negs = []
    orig_aug, orig_choice = crit.latent_augmentation, np.random.choice

    def rec_aug(features, labels):
        c0, a0 = int(labels[0, 0]), int(labels[0, 1])
        f, l, attr_id = orig_aug(features, labels)
        plan_on[c0 * A + a0] = attr_id > 0
        plan_attr[c0 * A + a0] = max(attr_id - 1, 0)
        return f, l, attr_id

    def rec_choice(cands, size):
        r = orig_choice(cands, size)
        negs.append(r.copy())
        return r
    crit.latent_augmentation = rec_aug
    cuda_float = getattr(torch.cuda, "FloatTensor", None)
    fwd_attr = RP.AttributeFittingModel.forward
    try:
        torch.cuda.FloatTensor = torch.FloatTensor                     # fixture generation only: the reference allocates on "cuda"
        np.random.choice = rec_choice
        random.seed(3)
        np.random.seed(4)
        F = F0.clone()
        labels = labels0.clone()
        with torch.no_grad():
            loss, pos_loss, neg_loss = crit(F, labels, T3)
    finally:
        np.random.choice = orig_choice
        torch.cuda.FloatTensor = cuda_float
    neg = torch.zeros(N, K, dtype=torch.long)
    uts = [int(c) for c in torch.unique(cat) if int(c) != -1]
    assert len(uts) == len(negs)
    for c, r in zip(uts, negs):
        neg[cat == c] = torch.from_numpy(r)
    sd = crit.projection_model.state_dict()
    out.update({"lat_F": F0.numpy(), "lat_T": T3.numpy(), "lat_labels": labels0.numpy(), "lat_neg": neg.numpy(),
                "lat_augment_categories": crit.augment_categories.numpy(), "lat_plan_on": plan_on, "lat_plan_attr": plan_attr,
                "lat_F_after": F.numpy(), "lat_labels_after": labels.numpy(), "lat_loss": loss.reshape(1).numpy(),
                "lat_pos_loss": pos_loss.numpy(), "lat_neg_loss": neg_loss.numpy(),
                "lat_proj_w": torch.stack([sd["attr_linears.%d.weight" % i] for i in range(8)]).numpy(),
                "lat_proj_b": torch.stack([sd["attr_linears.%d.bias" % i] for i in range(8)]).numpy()})
    np.savez_compressed(os.path.join(HERE, "contrastive_ce_latent.npz"), **out)
    print("CE + latent fixture: ce losses", {k: float(v[0]) for k, v in out.items() if k.endswith("_loss") and k.startswith("ce_")},
          "latent: %d of %d targets augmented, %d rows changed, loss %.5f" % (int(plan_on.sum()), len(uts),
                                                                              int((F != F0).any(1).sum()), float(loss)))


def _loss_utils():
    if "torchmetrics" not in sys.modules:
        sys.modules["torchmetrics"] = types.SimpleNamespace(Metric=object)
    import lib.losses.utils as U
    return U


def feature_sim_fixture():
    U = _loss_utils()
    fx = np.load(os.path.join(HERE, "contrastive_loss.npz"))
    cfg = types.SimpleNamespace(representation_distance_type="cos")
    out = {}
    for tag in ("c512", "c96"):
        F, T = torch.from_numpy(fx[tag + "_F"]), torch.from_numpy(fx[tag + "_T"])
        sim = U.feature_sim(F.clone(), T, cfg)
        # 3-D anchors (category, attribute, C): the reference keeps attribute 0 (:83-84)
        T3 = torch.stack([T, T.flip(0), T * 0.5], 1)
        sim3 = U.feature_sim(F.clone(), T3, cfg)
        assert torch.equal(sim, sim3)
        out[tag + "_sim"] = sim.numpy().astype(np.float32)
        out[tag + "_pred"] = sim.argmax(1).numpy()
    np.savez_compressed(os.path.join(HERE, "feature_sim.npz"), **out)
    print("feature_sim fixture:", {k: v.shape for k, v in out.items()})


def balancing_fixture():
    U = _loss_utils()
    L, n = 20, 5000
    g = torch.Generator().manual_seed(11)
    targets = torch.randint(-1, L, (n,), generator=g)
    targets[targets == 17] = 3                                   # one class absent from the batch
    foc = torch.zeros(L, 3, dtype=torch.bool)
    foc[:7, 0] = True; foc[7:14, 1] = True; foc[14:, 2] = True
    loss = torch.rand(n, generator=g) + 0.1
    ds = types.SimpleNamespace(NUM_LABELS=L, frequency_organized_cats=foc)
    out = {"targets": targets.numpy(), "foc": foc.numpy(), "loss": loss.numpy()}
    for tag, (hr, cr) in {"a": (0.3, 0.6), "b": (0.0, 0.45)}.items():
        cfg = types.SimpleNamespace(ignore_label=-1, balanced_sample_head_ratio=hr, balanced_sample_common_ratio=cr)
        keep = np.zeros(L, np.int64)
        for c in range(L):                                       # one-hot loss: mean * n = points of class c that were kept
            np.random.seed(100 + c)
            m, _, _ = U.sample_categories_for_balancing((targets == c).float(), cfg, ds, targets)
            keep[c] = int(round(float(m) * n))
        np.random.seed(5)
        m, (head, common, tail), items = U.sample_categories_for_balancing(loss.clone(), cfg, ds, targets)
        out.update({tag + "_ratios": np.array([hr, cr], np.float64), tag + "_keep": keep, tag + "_head": head.numpy(),
                    tag + "_common": common.numpy(), tag + "_tail": tail.numpy(), tag + "_items": items.numpy()})
    np.savez_compressed(os.path.join(HERE, "balancing.npz"), **out)
    print("balancing fixture: keep(a) =", out["a_keep"].tolist(), "keep(b) =", out["b_keep"].tolist())


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


def insseg_fixture(seed=11, n=1400):
    """downstream/insseg: the reference's instance-seg model (insseg_models/insseg_res16unet.py) through the alias
    package on the oracle backend -> state-dict manifest + (offsets, logits) on a small scene + the offset losses of
    lib/pl_Trainer.py:271-299 restated with the reference's own expressions."""
    sys.path.insert(0, "/root/reference/downstream/insseg")
    from insseg_models import insseg_res16unet as M
    cfg = types.SimpleNamespace(optimizer=types.SimpleNamespace(bn_momentum=0.02), net=types.SimpleNamespace(conv1_kernel_size=3))
    man = {}
    for name in ("Res16UNet14A", "Res16UNet34C"):
        m = getattr(M, name)(3, 20, cfg)
        man[name] = {"num_parameters": int(sum(p.numel() for p in m.parameters())),
                     "state_dict": [[k, list(v.shape)] for k, v in m.state_dict().items()]}
    prev = ME.set_backend(OracleBackend("c"))
    try:
        torch.manual_seed(0)
        m = deterministic_init(M.Res16UNet14A(3, 20, cfg), 42)
        m.train()
        coords = small_scene(seed, n=n)
        rng = np.random.default_rng(seed)
        feats = rng.uniform(-0.5, 0.5, (coords.shape[0], 3)).astype(np.float32)
        x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords))
        pt_offsets, soutput, out_feats = m(x)
        # losses exactly as pl_Trainer.py:283-299 writes them
        inst = rng.integers(-1, 6, coords.shape[0])
        centers = np.zeros((coords.shape[0], 3), np.float32)
        for i in range(6):
            sel = inst == i
            if sel.any():
                centers[sel] = coords[sel, 1:].mean(0)
        VOXEL = 0.02
        gt_offsets = torch.from_numpy(centers) - torch.from_numpy(coords[:, 1:]).float()
        gt_offsets *= VOXEL
        pt_diff = pt_offsets.F - gt_offsets
        pt_dist = torch.sum(torch.abs(pt_diff), dim=-1)
        valid = (torch.from_numpy(inst) != -1).float()
        offset_norm_loss = torch.sum(pt_dist * valid) / (torch.sum(valid) + 1e-6)
        gt_offsets_norm = torch.norm(gt_offsets, p=2, dim=1)
        gt_offsets_ = gt_offsets / (gt_offsets_norm.unsqueeze(-1) + 1e-8)
        pt_offsets_norm = torch.norm(pt_offsets.F, p=2, dim=1)
        pt_offsets_ = pt_offsets.F / (pt_offsets_norm.unsqueeze(-1) + 1e-8)
        direction_diff = - (gt_offsets_ * pt_offsets_).sum(-1)
        offset_dir_loss = torch.sum(direction_diff * valid) / (torch.sum(valid) + 1e-6)
        np.savez_compressed(os.path.join(HERE, "insseg_res16unet14a_forward.npz"), coords=coords, feats=feats,
                            offsets=pt_offsets.F.detach().numpy(), logits=soutput.F.detach().numpy(), inst=inst, centers=centers,
                            voxel=np.float32(VOXEL), norm_loss=np.float32(offset_norm_loss.item()),
                            dir_loss=np.float32(offset_dir_loss.item()))
        print("insseg fixture:", coords.shape, pt_offsets.F.shape, float(offset_norm_loss), float(offset_dir_loss))
    finally:
        ME.set_backend(prev)
    with open(os.path.join(HERE, "insseg_manifest.json"), "w") as f:
        json.dump(man, f)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "insseg":
        insseg_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "distances":
        distance_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ce_latent":
        ce_and_latent_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "losses":
        feature_sim_fixture()
        balancing_fixture()
        sys.exit(0)
    manifest()
    contrastive()
    distance_fixture()
    ce_and_latent_fixture()
    feature_sim_fixture()
    balancing_fixture()
    forward_fixture("Res16UNet14A", 3, 1500)
    forward_fixture("Res16UNet34C", 5, 1200)
