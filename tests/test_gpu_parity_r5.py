"""Round-5 parity tests on the GPU: the rest of the reference's loss file on the engine
(/root/reference/lib/losses/ContrastiveLanguageLoss.py:62-70,160-165,196-237, models/projection_models.py:4-19)."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.parity("reference-generated golden fixtures loaded at module level")]
DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")
FX = np.load(os.path.join(G, "contrastive_loss.npz"))
CL = np.load(os.path.join(G, "contrastive_ce_latent.npz"))


@pytest.mark.parametrize("dtype,tol_loss,tol_grad", [(torch.float32, 5e-6, 2e-6), (torch.bfloat16, 2e-2, None)])
@pytest.mark.parametrize("tag", ["c512", "c96"])
def test_contrastive_ce_loss_on_the_engine_matches_reference_fixture(tag, dtype, tol_loss, tol_grad):
    """`embedding_loss_type=contrast_ce`: lgs_clip_similarity (MFMA contraction) + lgs_ce_forward_backward as one autograd node
    against the reference class's own forward / backward (fixture generated on CPU by tests/golden/make_fixtures.py ce_latent)"""
    from languagegroundedsemseg_amd import engine
    from languagegroundedsemseg_amd.losses import ReferenceContrastiveLanguageCELoss
    from test_losses_cpu import _RefConfig
    g = lambda k: torch.from_numpy(FX["%s_%s" % (tag, k)]).to(DEV)
    crit = ReferenceContrastiveLanguageCELoss(_RefConfig(), 200).to(DEV)
    F = g("F").to(dtype).requires_grad_(True)
    engine.dispatch_counts(reset=True)
    loss, z, _ = crit(F, g("labels"), g("T"))
    loss.backward()
    sites = engine.dispatch_counts()
    assert any(k.startswith("k_ce_fwd_bwd") for k in sites) and any("k_conv_gather" in k or "k_row_invnorm" in k for k in sites), sites
    assert abs(float(loss) - float(CL["ce_cos_%s_loss" % tag][0])) < tol_loss
    want = torch.from_numpy(CL["ce_cos_%s_gradF" % tag]).to(DEV)
    if tol_grad is not None:
        assert torch.allclose(F.grad.float(), want, atol=tol_grad, rtol=1e-3)
    else:
        rel = float((F.grad.float() - want).norm() / want.norm())
        assert rel < 2e-2, rel


def test_latent_attribute_augmentation_on_the_device_matches_reference_fixture():
    from test_losses_cpu import _latent_case
    crit, plan, t = _latent_case(DEV)
    F, labels = t("F").clone(), t("labels").clone()
    loss, pos, neg = crit(F, labels, t("T"), neg_indices=t("neg"), aug_plan=plan)
    assert torch.allclose(F, t("F_after"), atol=5e-6) and torch.equal(labels, t("labels_after"))
    assert torch.allclose(pos, t("pos_loss"), atol=5e-6) and torch.allclose(neg, t("neg_loss"), atol=5e-6)
    # and through autograd: the projected rows carry their gradient back through the attribute's linear map
    F2 = t("F").clone().requires_grad_(True)
    Fw = F2 * 1.0
    l2, _, _ = crit(Fw, t("labels").clone(), t("T"), neg_indices=t("neg"), aug_plan=plan)
    l2.sum().backward()
    assert torch.isfinite(F2.grad).all() and float(F2.grad.abs().sum()) > 0


def test_syncbn_record_combination_matches_full_batch_statistics():
    """lgs_bn_stats ([mean | M2 | count] per rank) + lgs_bn_sync_combine (Chan's parallel combination, running statistics,
    1 / N) over three unequal "ranks" of one batch, in process, against torch's statistics of the concatenated batch
    (/root/reference/main.py:121-123: MinkowskiSyncBatchNorm) -- the kernels every N > 1 step runs, for which the dispatch-coverage
    assertion found only self-comparison tests"""
    import MinkowskiEngine as ME
    be = ME.get_backend()
    torch.manual_seed(4)
    c = 96
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-5)):
        parts = [(torch.randn(n, c, device=DEV) * s + m).to(dtype) for n, s, m in ((5000, 2.0, 1.0), (1234, 0.5, -3.0), (40000, 1.0, 0.2))]
        recs = torch.stack([be.bn_stats(p) for p in parts])
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        stats, inv_n = be.bn_sync_combine(recs, c, 1e-5, 0.1, rm, rv, nbt)
        full = torch.cat(parts).double()
        mean, var = full.mean(0), full.var(0, unbiased=False)
        assert torch.allclose(stats[:c].double(), mean, atol=tol, rtol=1e-5)
        assert torch.allclose(stats[c:].double(), 1.0 / torch.sqrt(var + 1e-5), atol=tol, rtol=1e-4)
        assert abs(float(inv_n) - 1.0 / full.shape[0]) < 1e-12 and int(nbt) == 1
        assert torch.allclose(rm.double(), 0.1 * mean, atol=tol) and torch.allclose(rv.double(), 0.9 + 0.1 * full.var(0, unbiased=True), rtol=1e-4)


@pytest.mark.parametrize("cin,cout,ks", [(96, 96, 3), (32, 64, 2), (128, 96, 3), (96, 200, 1), (256, 256, 3), (36, 20, 3)])
def test_fp32_weight_gradient_staged_through_lds_is_the_pairwise_kernel_bit_for_bit(cin, cout, ks):
    """k_wgrad_f32_lds (16-byte row loads into LDS, round 5) issues k_wgrad_f32's MFMA sequence on the same compacted pairs: the
    gradient must be IDENTICAL, on 3^3, strided 2^3 (forward and transposed use) and 1x1 maps, and match the oracle"""
    import MinkowskiEngine as ME
    from helpers import small_scene
    from languagegroundedsemseg_amd import engine
    from oracle import oracle as orc
    coords = small_scene(11, n=6000, extent=40)
    c = torch.from_numpy(coords).to(DEV)
    x = ME.SparseTensor(torch.zeros(coords.shape[0], 1, device=DEV), c)
    mgr, k0 = x.coordinate_manager, x.coordinate_map_key
    k1 = mgr.stride(k0, 2) if ks == 2 else k0
    km = mgr.kernel_map_handle(k0, k1, ks)
    for transposed in ((False, True) if ks == 2 else (False,)):
        n_in, n_out = km._rows(transposed)
        g = torch.Generator(device=DEV).manual_seed(3)
        a = torch.randn(n_in, cin, device=DEV, generator=g)
        b = torch.randn(n_out, cout, device=DEV, generator=g)
        engine.dispatch_counts(reset=True)
        with engine.tuning(WGRAD_F32_LDS=1):
            w1 = km.conv_wgrad(a, b, transposed)
        sites = engine.dispatch_counts(reset=True)
        assert any(s.startswith("k_wgrad_f32_lds") for s in sites), sites
        with engine.tuning(WGRAD_F32_LDS=0):
            w0 = km.conv_wgrad(a, b, transposed)
        assert torch.equal(w0, w1)
        # round 6, the default (WGRAD_F32_LDS=2): the same staged rows as three exactly-split bf16 planes, six bf16 products per
        # operand pair -- not bit-identical (another summation order, terms below 2^-24 dropped), held to the same 2e-5 of the oracle
        if cin % 4 == 0 and cout % 4 == 0:
            w2 = km.conv_wgrad(a, b, transposed)
            assert any(s.startswith("k_wgrad_f32s_lds") for s in engine.dispatch_counts(reset=True))
            assert float((w2.double() - w1.double()).norm() / w1.double().norm()) < 1e-5
        kk, ii, oo = (t.cpu().numpy() for t in km.export())
        if transposed:
            ii, oo = oo, ii
        want = orc.conv_wgrad(a.cpu().numpy(), b.cpu().numpy(), (kk, ii, oo), ks ** 3)
        rel = float(np.linalg.norm(w1.cpu().numpy().reshape(want.shape) - want) / np.linalg.norm(want))
        assert rel < 2e-5, rel
        if cin % 4 == 0 and cout % 4 == 0:
            rel2 = float(np.linalg.norm(w2.cpu().numpy().reshape(want.shape) - want) / np.linalg.norm(want))
            assert rel2 < 2e-5, rel2
