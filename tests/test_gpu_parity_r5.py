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
