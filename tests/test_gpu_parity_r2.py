"""Round-2 parity tests on the GPU (VERDICT r01 "next round" item 1):
  * the BENCHMARKED path -- Res16UNet34C with bf16 feature storage -- forward + backward against the fp32 CPU oracle,
    on the reference-generated 34C fixture scene and on a >= 60k-voxel synthetic 2 cm scene (big-tile conv configs,
    slot split, 4096-position wgrad ranges inside a whole network), bounds = measured deviation + margin;
  * BASELINE configs[2]: Res16UNet34D (512-d) + the CLIP text-anchor loss, fp32 within 1e-3 and bf16, vs the oracle;
  * the fused CLIP loss kernels (lgs_clip_loss_forward / _backward) vs the reference's own feat_dist golden and vs
    autograd through the reference formulation; feature_sim + argmax vs the reference's feature_sim fixture;
  * the loss-side sampling (balanced sampling fixture, both negative-sampling modes) on HIP tensors;
  * cross-entropy on class counts that are not a multiple of the 16-byte width (20 ScanNet classes in bf16)."""
import os

import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from helpers import Cfg, deterministic_init
from languagegroundedsemseg_amd.models import load_model
from oracle.backend import OracleBackend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(1e-30, np.linalg.norm(b.astype(np.float64))))


def ce_step(name, coords, feats, labels, device, dtype, n_classes=20):
    m = deterministic_init(load_model(name)(3, n_classes, Cfg()), 42).to(device).train()
    x = ME.SparseTensor(torch.from_numpy(feats).to(device).to(dtype), torch.from_numpy(coords).to(device))
    logits, _ = m(x)
    loss = torch.nn.functional.cross_entropy(logits.F.float(), torch.from_numpy(labels).to(device), ignore_index=-1)
    loss.backward()
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters()}
    return logits.F.detach().float().cpu().numpy(), float(loss), grads


def on_oracle(fn, impl):
    prev = ME.set_backend(OracleBackend(impl))
    try:
        return fn()
    finally:
        ME.set_backend(prev)


def grad_report(h, o, tag):
    errs = {k: rel_l2(h[k], o[k]) for k in o}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    tot = float(np.sqrt(sum(np.linalg.norm(h[k].astype(np.float64) - o[k]) ** 2 for k in o) /
                        sum(np.linalg.norm(o[k].astype(np.float64)) ** 2 for k in o)))
    print("%s: gradient rel-L2 over all parameters %.3e, median tensor %.3e, worst %s" % (
        tag, tot, float(np.median(list(errs.values()))), ", ".join("%s=%.3e" % kv for kv in worst)))
    return errs, tot


# ------------------------------------------------------------------------------------------- 34C, the benchmarked path
# How bf16 storage is judged.  bf16 activations perturb every layer's output by ~2^-9; through ~70 conv/BN/ReLU layers
# that moves the logits by ~2 % and -- because BatchNorm's backward subtracts the batch means and ReLU gates flip -- the
# parameter gradients by tens of percent of their (small) norm.  That is a property of bf16 storage, not of the kernels:
# the CPU ORACLE run with bf16-rounded activations (its own conv / BN / ReLU code, rounding after every op) deviates from
# its fp32 run by the same amount.  So the HIP bf16 step is held (a) to absolute bounds on logits / loss and (b) to the
# oracle's own bf16 noise floor for the gradients: a wrong tile config or a dropped kernel-map entry shows up as a
# deviation ABOVE that floor (and in the per-op bf16 tests of test_gpu_engine.py, which hold 2e-2 per op).
def structured_labels(coords, n_classes=20):
    """labels that depend on the geometry (a learnable signal), 10 % ignored"""
    c = coords.astype(np.int64)
    lab = ((c[:, 1] // 16) + 3 * (c[:, 2] // 16) + 5 * (c[:, 3] // 8)) % n_classes
    lab[np.random.default_rng(0).random(c.shape[0]) < 0.1] = -1
    return lab


def check_bf16_against_noise_floor(coords, feats, labels, impl, tag):
    h_logits, h_loss, h_g = ce_step("Res16UNet34C", coords, feats, labels, DEV, torch.bfloat16)
    o_logits, o_loss, o_g = on_oracle(lambda: ce_step("Res16UNet34C", coords, feats, labels, "cpu", torch.float32), impl)
    b_logits, b_loss, b_g = on_oracle(lambda: ce_step("Res16UNet34C", coords, feats, labels, "cpu", torch.bfloat16), impl)
    e, eb = rel_l2(h_logits, o_logits), rel_l2(b_logits, o_logits)
    print("34C bf16 storage, %s (%d voxels): logit rel-L2 HIP %.3e / bf16-storage oracle %.3e, loss %.5f / %.5f vs fp32 oracle %.5f" % (
        tag, coords.shape[0], e, eb, h_loss, b_loss, o_loss))
    errs, tot = grad_report(h_g, o_g, "34C bf16 HIP vs fp32 oracle, " + tag)
    berrs, btot = grad_report(b_g, o_g, "34C bf16-storage ORACLE vs fp32 oracle (noise floor), " + tag)
    assert e < 4e-2 and e < 1.5 * eb + 5e-3                 # measured 1.8e-2 / 2.7e-2 (HIP) vs 1.6e-2 (oracle bf16)
    assert abs(h_loss - o_loss) < 5e-3                      # measured 8e-4 / 2e-4
    assert tot < 1.25 * btot + 0.02                         # measured 0.456 vs 0.415 on the fixture scene
    assert float(np.median(list(errs.values()))) < 1.25 * float(np.median(list(berrs.values()))) + 0.02
    return o_logits


def test_res16unet34c_bf16_on_fixture_scene_vs_fp32_oracle():
    fx = np.load(os.path.join(G, "res16unet34c_forward.npz"))
    coords, feats = fx["coords"], fx["feats"]
    labels = np.random.default_rng(0).integers(-1, 20, coords.shape[0]).astype(np.int64)
    o_logits = check_bf16_against_noise_floor(coords, feats, labels, "c", "fixture scene")
    assert np.abs(o_logits - fx["logits"]).max() < 1e-4            # the oracle run IS the reference-generated fixture


def test_res16unet34c_bf16_on_60k_voxel_scene_vs_fp32_oracle():
    """the big-tile gather configs, the slot split and the 4096-position wgrad ranges inside a whole network"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, feats, _ = make_batch([7], voxel=0.02, n_target=70000)
    assert coords.shape[0] >= 65536 - 255                          # n_pad >= 65536: the "big map" tile configurations
    check_bf16_against_noise_floor(coords, feats, structured_labels(coords), "torch", "70k-voxel 2cm scene")


def test_res16unet34c_fp32_on_60k_voxel_scene_within_1e3_of_oracle():
    """north_star bar (logits within 1e-3 fp32) at a size where every big-map code path is live"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, feats, _ = make_batch([7], voxel=0.02, n_target=70000)
    labels = structured_labels(coords)
    h_logits, h_loss, h_g = ce_step("Res16UNet34C", coords, feats, labels, DEV, torch.float32)
    o_logits, o_loss, o_g = on_oracle(lambda: ce_step("Res16UNet34C", coords, feats, labels, "cpu", torch.float32), "torch")
    d = float(np.abs(h_logits - o_logits).max())
    print("34C fp32, %d voxels: max |dlogit| %.3e" % (coords.shape[0], d))
    errs, tot = grad_report(h_g, o_g, "34C fp32 70k")
    assert d < 1e-3                                          # measured 1.0e-4
    assert abs(h_loss - o_loss) < 1e-4
    assert tot < 1e-2                                        # measured 2.8e-3 (both sides fp32 with different summation orders)


# ------------------------------------------------------------------------------------------- 34D + CLIP loss (configs[2])
def clip_step(coords, feats, labels, anchors, neg, device, dtype):
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
    crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
    m = deterministic_init(load_model("Res16UNet34D")(3, 20, Cfg()), 42).to(device).train()
    m.representation_only(True)
    x = ME.SparseTensor(torch.from_numpy(feats).to(device).to(dtype), torch.from_numpy(coords).to(device))
    out = m(x)
    loss, pos, ngl = crit(out.F, torch.from_numpy(labels).to(device), torch.from_numpy(anchors).to(device),
                          neg_indices=neg.to(device))
    loss.backward()
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    return float(loss), out.F.detach().float().cpu().numpy(), grads


@pytest.mark.parametrize("size", ["fixture", "70k"])
def test_res16unet34d_clip_step_vs_oracle(size):
    """scripts/text_representation_train.sh:7 -> Res16UNet34D (models/clip_models.py:205-215): 512-d representation,
    ReLU-free last block, fused CLIP loss; fp32 within the north-star 1e-3, bf16 reported against the fp32 oracle.
    The 70k scene runs the wide-channel tiles (256 x 256 workgroup tile, 16 column blocks) inside the network."""
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
    from languagegroundedsemseg_amd.synthetic import make_batch, text_anchors
    if size == "fixture":
        fx = np.load(os.path.join(G, "res16unet34c_forward.npz"))
        coords, feats = fx["coords"], fx["feats"]
    else:
        coords, feats, _ = make_batch([9], voxel=0.02, n_target=70000)
    rng = np.random.default_rng(2)
    labels = rng.integers(-1, 200, coords.shape[0]).astype(np.int64)
    anchors = text_anchors(200, 512)
    neg = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3).sample_negatives(
        torch.from_numpy(labels), generator=torch.Generator().manual_seed(5))
    o = on_oracle(lambda: clip_step(coords, feats, labels, anchors, neg, "cpu", torch.float32), "c" if size == "fixture" else "torch")
    h = clip_step(coords, feats, labels, anchors, neg, DEV, torch.float32)
    d = float(np.abs(h[1] - o[1]).max())
    print("34D fp32 %s (%d voxels): loss %.6f vs %.6f, max |dfeature| %.3e (feature scale %.3f)" % (
        size, coords.shape[0], h[0], o[0], d, float(np.abs(o[1]).max())))
    errs, tot = grad_report(h[2], o[2], "34D fp32 " + size)
    assert abs(h[0] - o[0]) < 1e-4
    assert d < 1e-3                                          # measured 8e-5 / 3e-4 at a feature scale of 25 / 40
    assert tot < 1.5e-2                                      # measured 4.5e-3 / 5.2e-3
    b = clip_step(coords, feats, labels, anchors, neg, DEV, torch.bfloat16)
    e = rel_l2(b[1], o[1])
    print("34D bf16 %s: loss %.6f vs %.6f, feature rel-L2 %.3e" % (size, b[0], o[0], e))
    errs, tot = grad_report(b[2], o[2], "34D bf16 HIP vs fp32 oracle, " + size)
    # gradients: against the bf16-storage oracle's own deviation (see the note above the 34C tests)
    if size == "fixture":
        ob = on_oracle(lambda: clip_step(coords, feats, labels, anchors, neg, "cpu", torch.bfloat16), "c")
        berrs, btot = grad_report(ob[2], o[2], "34D bf16-storage ORACLE vs fp32 oracle (noise floor), " + size)
        eb = rel_l2(ob[1], o[1])
        print("34D bf16-storage oracle: feature rel-L2 %.3e" % eb)
    else:
        # the 512-channel CPU oracle on 70 k voxels takes minutes per pass: the bf16-STORAGE pass of the oracle for exactly
        # this scene / seeds / weights is cached in tests/golden/noise_floor_34d_70k.npz (written by
        # tests/golden/make_noise_floor.py, which runs both oracle passes); the fp32 pass above is still run here, and its
        # loss must be the one the cache was generated with
        nf = np.load(os.path.join(G, "noise_floor_34d_70k.npz"))
        assert int(nf["n_voxels"]) == coords.shape[0] and abs(float(nf["loss_fp32"]) - o[0]) < 1e-5, \
            "noise_floor_34d_70k.npz does not belong to this scene / model: re-run tests/golden/make_noise_floor.py"
        rows = nf["feature_sample_rows"]
        assert np.abs(o[1][rows] - nf["feature_sample_fp32"]).max() < 1e-3      # same oracle, another host's BLAS threading
        btot, eb = float(nf["grad_rel_l2_total"]), float(nf["feature_rel_l2"])
        print("34D bf16-storage ORACLE noise floor (cached): gradient rel-L2 %.4f, feature rel-L2 %.3e" % (btot, eb))
    assert abs(b[0] - o[0]) < 5e-3                           # measured 1.6e-4
    assert e < 6e-2 and e < 1.5 * eb + 5e-3                  # measured 3.3e-2 / 4.3e-2
    assert tot < 1.25 * btot + 0.02


# ------------------------------------------------------------------------------------------- fused CLIP loss kernels
def ref_clip_loss(F, T, labels, neg):
    """the reference formulation (ContrastiveLanguageLoss.py:73-95,185-192) in float64 with autograd"""
    F = F.double().clone().requires_grad_(True)
    fn = torch.nn.functional.normalize(F, dim=1)
    tn = torch.nn.functional.normalize(T.double(), dim=1)
    valid = labels != -1
    lab = labels.clamp_min(0)
    dp = torch.where(valid, 1 - (fn * tn[lab]).sum(1), torch.zeros((), dtype=torch.float64))
    dn = torch.where(valid, 1 - torch.einsum("nc,nkc->nk", fn, tn[neg]).mean(1), torch.zeros((), dtype=torch.float64))
    sim = fn @ tn.t()
    return F, dp, dn, sim


@pytest.mark.parity("the reference formulation restated in torch (float64 where stated)")
@pytest.mark.parametrize("c,na,k,n", [(512, 200, 3, 257), (96, 200, 3, 1000), (64, 20, 1, 130), (32, 64, 7, 4099),
                                      (128, 100, 2, 511), (256, 224, 3, 129)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_clip_loss_kernels_match_reference_formulation(c, na, k, n, dtype):
    be = ME.get_backend()
    g = torch.Generator().manual_seed(c + na + n)
    F = (torch.randn(n, c, generator=g) * torch.rand(n, 1, generator=g).add(0.1)).to(dtype).float()   # rows of very different norm
    T = torch.randn(na, c, generator=g)
    labels = torch.randint(0, na, (n,), generator=g)
    labels[torch.rand(n, generator=g) < 0.15] = -1
    neg = torch.randint(0, na, (n, k), generator=g)
    gp, gn = torch.randn(n, generator=g), torch.randn(n, generator=g)
    d_pos, d_neg, pred, saved, sim = be.clip_loss_forward(F.to(DEV).to(dtype), T.to(DEV), labels.to(DEV), neg.to(DEV), -1, want_sim=True)
    Fr, dp, dn, sr = ref_clip_loss(F, T, labels, neg)
    tol = 2e-6 if dtype == torch.float32 else 6e-3     # bf16: the anchors are rounded to bf16 for the MFMA
    assert float((d_pos.cpu().double() - dp).abs().max()) < tol
    assert float((d_neg.cpu().double() - dn).abs().max()) < tol
    assert float((sim.cpu().double() - sr).abs().max()) < tol
    inv = saved[4].cpu().double()
    assert float((inv * F.double().norm(dim=1) - 1).abs().max()) < 1e-5
    # arg-max: equal wherever the top two similarities are separated by more than the arithmetic tolerance
    top2 = sr.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 4 * tol
    assert torch.equal(pred.cpu()[clear], sr.argmax(1)[clear]) and int(clear.sum()) > (n // 2 if dtype == torch.float32 else 10)
    assert torch.equal(pred.cpu(), sim.cpu().argmax(1))             # and always consistent with its own similarity matrix
    # backward against autograd through the reference formulation
    (dp * gp.double() + dn * gn.double()).sum().backward()
    gf = be.clip_loss_backward(saved, d_pos, d_neg, gp.to(DEV), gn.to(DEV), -1)
    e = rel_l2(gf.float().cpu().numpy(), Fr.grad.float().numpy())
    assert e < (1e-5 if dtype == torch.float32 else 8e-3), e
    assert float(gf.float().cpu()[labels == -1].abs().max()) == 0.0


def test_contrastive_loss_module_runs_fused_and_matches_golden_incl_gradient():
    """the module-level path (ContrastiveLanguageLoss.forward -> _ClipLossFused) vs the reference's feat_dist golden, and
    its gradient vs the dense formulation through autograd"""
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss, clip_similarity
    fx = np.load(os.path.join(G, "contrastive_loss.npz"))
    fs = np.load(os.path.join(G, "feature_sim.npz"))
    be = ME.get_backend()
    calls = []
    orig = be.clip_loss_forward
    be.clip_loss_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    for tag in ("c512", "c96"):
        g = lambda k: torch.from_numpy(fx["%s_%s" % (tag, k)]).to(DEV)
        crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
        F = g("F").clone().requires_grad_(True)
        loss, pos, neg, pred = crit(F, g("labels"), g("T"), neg_indices=g("neg"), return_pred=True)
        assert torch.allclose(pos, g("pos_loss"), atol=2e-6) and torch.allclose(neg, g("neg_loss"), atol=2e-6)
        assert abs(float(loss) - float(g("total"))) < 2e-6
        # pred == the reference's feature_sim(...).argmax(1) wherever the top-2 gap is resolvable in fp32
        sim_ref = torch.from_numpy(fs[tag + "_sim"])
        top2 = sim_ref.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert torch.equal(pred.cpu()[clear], torch.from_numpy(fs[tag + "_pred"])[clear]) and int(clear.sum()) > 250
        loss.backward()
        F2 = g("F").clone().requires_grad_(True)
        sim = clip_similarity(F2, g("T"))                                # dense path + torch gathers
        valid = g("labels") != -1
        lab = g("labels").clamp_min(0)
        dp = torch.where(valid, 1 - sim.gather(1, lab[:, None]).squeeze(1), torch.zeros((), device=DEV))
        dn = torch.where(valid, 1 - sim.gather(1, g("neg")).mean(1), torch.zeros((), device=DEV))
        (torch.relu(dp).mean() + torch.relu(0.6 - dn).mean()).backward()
        assert torch.allclose(F.grad, F2.grad, atol=1e-7, rtol=1e-4)
    del be.clip_loss_forward                                             # drop the instance-level spy
    assert len(calls) == 2, "the module must take the fused kernel path on the HIP backend"


def test_feature_sim_on_the_engine_matches_reference_fixture():
    """a12: feature_sim (lib/losses/utils.py:80-103, cosine) on lgs_clip_similarity vs the reference's output"""
    from languagegroundedsemseg_amd.losses import feature_sim, feature_sim_argmax
    fx = np.load(os.path.join(G, "contrastive_loss.npz"))
    fs = np.load(os.path.join(G, "feature_sim.npz"))
    for tag in ("c512", "c96"):
        F, T = torch.from_numpy(fx[tag + "_F"]).to(DEV), torch.from_numpy(fx[tag + "_T"]).to(DEV)
        sim = feature_sim(F, T)
        ref = torch.from_numpy(fs[tag + "_sim"])
        assert float((sim.cpu() - ref).abs().max()) < 2e-6
        top2 = ref.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert torch.equal(feature_sim_argmax(sim).cpu()[clear], torch.from_numpy(fs[tag + "_pred"])[clear])
        sim3 = feature_sim(F, torch.stack([T, T.flip(0)], 1))             # attribute anchors: attribute 0 is used
        assert torch.equal(sim3, sim)


# ------------------------------------------------------------------------------------------- loss-side sampling (8f-2)
def test_balanced_sampling_on_hip_tensors_matches_reference_fixture():
    from test_losses_cpu import check_balancing_against_reference_fixture
    check_balancing_against_reference_fixture(DEV)


def test_negative_sampling_modes_on_hip_tensors():
    """ContrastiveLanguageLoss.py:128-146 on the device: uniform over all other classes (clip_uniform_sampling) and
    uniform over the other classes PRESENT in the batch"""
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
    torch.manual_seed(1)
    cu = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3, uniform_sampling=True)
    lab = torch.randint(-1, 200, (200000,), device=DEV)
    neg = cu.sample_negatives(lab)
    assert neg.is_cuda and neg.shape == (200000, 3) and int(neg.min()) >= 0 and int(neg.max()) < 200
    v = lab >= 0
    assert not bool((neg[v] == lab[v][:, None]).any())
    hist = torch.bincount(neg[lab == 5].flatten(), minlength=200).float()
    assert hist[5] == 0 and float((hist / hist.sum() - 1 / 199).abs().max()) < 0.01
    cp = ContrastiveLanguageLoss(num_labels=50, num_negative_samples=4, uniform_sampling=False)
    present = torch.tensor([3, 17, 18, 40, 49], device=DEV)
    labels = present[torch.randint(0, 5, (20000,), device=DEV)]
    labels[::97] = -1
    neg = cp.sample_negatives(labels)
    v = labels != -1
    assert bool(torch.isin(neg[v], present).all()) and bool((neg[v] != labels[v][:, None]).all())
    for own in present.tolist():
        rows = neg[labels == own].flatten()
        freq = torch.stack([(rows == c).float().mean() for c in present.tolist() if c != own])
        assert float((freq - 0.25).abs().max()) < 0.02


# ------------------------------------------------------------------------------------------- cross-entropy, odd widths
@pytest.mark.parity("the reference formulation restated in torch (float64 where stated)")
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("classes", [20, 150, 13])
def test_cross_entropy_class_counts_off_the_16_byte_grid(classes, dtype, tol):
    """ScanNet-20 heads (downstream/insseg, lg_semseg configs): 20 bf16 logits are 40 bytes per row; out-of-range labels
    are ignored rows and must not be counted in the mean"""
    from languagegroundedsemseg_amd.losses import fused_cross_entropy
    torch.manual_seed(classes)
    x = (torch.randn(3001, classes) * 3).to(dtype).float()
    lab = torch.randint(-1, classes, (3001,))
    lab_h = lab.clone()
    lab_h[::50] = classes + 3                                          # out of range -> ignored
    lab_t = lab.clone()
    lab_t[::50] = -1
    xh = x.to(DEV).to(dtype).requires_grad_(True)
    loss = fused_cross_entropy(xh, lab_h.to(DEV), -1)
    (loss * 2.0).backward()
    xt = x.clone().requires_grad_(True)
    lt = torch.nn.functional.cross_entropy(xt, lab_t, ignore_index=-1)
    (lt * 2.0).backward()
    assert abs(float(loss) - float(lt)) < 1e-4
    assert rel_l2(xh.grad.float().cpu().numpy(), xt.grad.numpy()) < tol


# ------------------------------------------------------------------------------------------- BN statistics from the conv epilogue
@pytest.mark.parity("the reference formulation restated in torch (float64 where stated)")
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_epilogue_batchnorm_statistics_equal_the_column_reduction(dtype):
    """lgs_conv_forward(bn_partial) + lgs_bn_forward(conv_partials) == lgs_bn_forward reading the output itself: same
    mean / invstd / running statistics / normalised output, on 3^3 (big map: plain launch; small map: slot split -> no
    partials), strided 2^3 and transposed 2^3 (grouped view with padding groups) convolutions"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    be = ME.get_backend()
    coords, _, _ = make_batch([3], voxel=0.02, n_target=70000)
    c = torch.from_numpy(coords).to(DEV)
    x0 = ME.SparseTensor(torch.zeros(coords.shape[0], 3, device=DEV), c)
    m, k0 = x0.coordinate_manager, x0.coordinate_map_key
    k1 = m.stride(k0, 2)
    k2 = m.stride(k1, 2)
    k3 = m.stride(k2, 2)
    cases = [(m.kernel_map_handle(k0, k0, 3), False, m.size(k0), 32, 96, True),
             (m.kernel_map_handle(k0, k0, 3), False, m.size(k0), 64, 32, True),
             (m.kernel_map_handle(k0, k0, 3), False, m.size(k0), 128, 128, True),
             (m.kernel_map_handle(k0, k1, 2), False, m.size(k0), 32, 64, True),
             (m.kernel_map_handle(k0, k1, 2), True, m.size(k1), 64, 96, True),
             (m.kernel_map_handle(k0, k0, 1), False, m.size(k0), 128, 96, True),
             (m.kernel_map_handle(k3, k3, 3), False, m.size(k3), 128, 128, dtype == torch.float32)]   # bf16: slot split, no partials
    torch.manual_seed(0)
    for km, tr, n_in, cin, cout, expect in cases:
        f = (torch.randn(n_in, cin, device=DEV) + 0.3).to(dtype)
        w = torch.randn(km.K, cin, cout, device=DEV) * (1.0 / (cin * km.K) ** 0.5)
        pivot = torch.randn(cout, device=DEV) * 0.05
        out, cs = km.conv_forward(f, w, None, tr, bn_pivot=pivot, want_bn_stats=True)
        assert (cs is not None) == expect, (km.K, tr, cin, cout)
        if cs is None:
            continue
        from languagegroundedsemseg_amd import engine
        with engine.tuning(POINTWISE=0):      # (the same kernel without the epilogue: big fp32 1x1 launches otherwise take k_pointwise_f32)
            assert torch.equal(out, km.conv_forward(f, w, None, tr))   # the extra epilogue does not touch the output
        g, b = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.1
        rm1, rv1, rm2, rv2 = pivot.clone(), torch.ones(cout, device=DEV), pivot.clone(), torch.ones(cout, device=DEV)
        y1, s1 = be.bn_forward(out, g, b, 1e-5, 0.1, rm1, rv1, None, 1)
        y2, s2 = be.bn_forward(out, g, b, 1e-5, 0.1, rm2, rv2, None, 1, None, conv_stats=(cs[0], rm2))
        assert torch.allclose(s1, s2, rtol=2e-5, atol=1e-6), (km.K, tr, cin, cout, float((s1 - s2).abs().max()))
        assert torch.allclose(rm1, rm2, atol=1e-6) and torch.allclose(rv1, rv2, rtol=1e-5, atol=1e-7)
        assert float((y1.float() - y2.float()).abs().max()) <= (1e-4 if dtype == torch.float32 else 4e-2)


# ------------------------------------------------------------------------------------------- packed weight images
@pytest.mark.parametrize("optim", ["flat", "torch"])
def test_packed_weight_cache_never_serves_stale_weights(optim):
    """three training steps with the packed-image cache (one batched re-pack after FlatSGD.step(); version-stamped images
    under torch.optim.SGD) == the same steps packing on every call, bit for bit"""
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD
    from languagegroundedsemseg_amd.me.backend_hip import get_packed
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, feats, _ = make_batch([0, 1], voxel=0.05, n_target=6000)
    c, f = torch.from_numpy(coords).to(DEV), torch.from_numpy(feats).to(DEV).bfloat16()
    res = []
    for enabled in (True, False):
        get_packed().enabled = enabled
        try:
            m = deterministic_init(load_model("Res16UNet14A")(3, 20, Cfg()), 42).to(DEV).train()
            if optim == "flat":
                ddp = BucketedDDP(m, bucket_mb=1.0)
                opt = FlatSGD(ddp, lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-4)
            else:
                ddp, opt = None, torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
            for step in range(3):
                if ddp is not None:
                    ddp.zero_grad()
                else:
                    opt.zero_grad(set_to_none=True)
                logits, _ = m(ME.SparseTensor(f, c))
                logits.F.float().square().mean().backward()
                if ddp is not None:
                    ddp.finalize()
                opt.step()
            torch.cuda.synchronize()
            res.append(({k: p.detach().cpu().clone() for k, p in m.named_parameters()}, logits.F.detach().float().cpu()))
        finally:
            get_packed().enabled = True
    assert torch.equal(res[0][1], res[1][1])
    for k in res[0][0]:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    if optim == "flat":
        assert len(get_packed().entries) > 60 and get_packed().epoch >= 3


@pytest.mark.parity("the reference formulation restated in torch (float64 where stated)")
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_eval_mode_batchnorm_on_the_engine_matches_torch(dtype, tol):
    """MinkowskiBatchNorm in eval mode (running statistics) with the fused residual / ReLU tail, forward + backward"""
    torch.manual_seed(1)
    n, c = 5003, 96
    coords = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.arange(n, dtype=torch.int32)[:, None].repeat(1, 3)], 1).to(DEV)
    xs = ME.SparseTensor(torch.randn(n, c, device=DEV).to(dtype), coords)
    bn = ME.MinkowskiBatchNorm(c).to(DEV)
    with torch.no_grad():
        bn.bn.running_mean.uniform_(-0.5, 0.5); bn.bn.running_var.uniform_(0.5, 2.0)
        bn.bn.weight.uniform_(0.5, 1.5); bn.bn.bias.uniform_(-0.3, 0.3)
    bn.eval()
    for relu, with_res in ((False, False), (True, False), (True, True)):
        x = xs.F.detach().clone().requires_grad_(True)
        r = torch.randn(n, c, device=DEV).to(dtype).requires_grad_(True) if with_res else None
        xin = ME.SparseTensor(x, coordinate_map_key=xs.coordinate_map_key, coordinate_manager=xs.coordinate_manager)
        y = bn(xin, relu=relu, residual=r).F
        g = torch.randn_like(y)
        bn.zero_grad()
        y.backward(g)
        x2 = xs.F.detach().float().clone().requires_grad_(True)
        r2 = r.detach().float().clone().requires_grad_(True) if with_res else None
        t = torch.nn.functional.batch_norm(x2, bn.bn.running_mean, bn.bn.running_var, bn.bn.weight, bn.bn.bias, False, 0.0, bn.bn.eps)
        if with_res:
            t = t + r2
        if relu:
            t = torch.relu(t)
        gw0, gb0 = bn.bn.weight.grad.clone(), bn.bn.bias.grad.clone()
        bn.zero_grad()
        t.backward(g.float())
        assert rel_l2(y.detach().float().cpu().numpy(), t.detach().cpu().numpy()) < tol
        assert rel_l2(x.grad.float().cpu().numpy(), x2.grad.cpu().numpy()) < tol
        if with_res:
            assert rel_l2(r.grad.float().cpu().numpy(), r2.grad.cpu().numpy()) < tol
        assert rel_l2(gw0.cpu().numpy(), bn.bn.weight.grad.cpu().numpy()) < tol and rel_l2(gb0.cpu().numpy(), bn.bn.bias.grad.cpu().numpy()) < tol


# ------------------------------------------------------------------------------------------- zero-copy ME.cat
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_zero_copy_cat_equals_the_copying_cat(dtype):
    """ME.cat(up, skip) without torch.cat: the norm that produces `up` writes straight into the left-hand columns of the concat
    buffer and the skip half is copied in beside it once (me/deferred.py); the backward hands out column slices of the gradient
    that the norms read through a row stride: logits and every gradient must be bit-identical to the torch.cat path"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    be = ME.get_backend()
    coords, feats, _ = make_batch([0, 1], voxel=0.05, n_target=8000)
    c, f = torch.from_numpy(coords).to(DEV), torch.from_numpy(feats).to(DEV).to(dtype)
    lab = torch.from_numpy(np.random.default_rng(0).integers(-1, 20, coords.shape[0]).astype(np.int64)).to(DEV)
    res = []
    calls = {"cat": 0}
    orig_cat = torch.cat

    def counting_cat(*a, **k):
        calls["cat"] += 1
        return orig_cat(*a, **k)
    for zero_copy in (True, False):
        be.bn_out_into = zero_copy
        try:
            m = deterministic_init(load_model("Res16UNet14A")(3, 20, Cfg()), 42).to(DEV).train()
            calls["cat"] = 0
            torch.cat = counting_cat
            try:
                logits, fmap = m(ME.SparseTensor(f, c))
                logits.F                                        # (values drive execution: the tail of the record runs here)
            finally:
                torch.cat = orig_cat
            assert calls["cat"] == (0 if zero_copy else 4), calls
            torch.nn.functional.cross_entropy(logits.F.float(), lab, ignore_index=-1).backward()
            res.append((logits.F.detach().float().cpu(), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}))
        finally:
            be.bn_out_into = True
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
