"""Whole-network parity on the GPU: Res16UNet forward + backward on the HIP engine against the CPU
oracle (BASELINE config[0]-style scene sizes that the oracle finishes in seconds) and against the
committed reference-generated fixtures."""
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


def run_model(name, coords, feats, labels, device, dtype=torch.float32):
    m = deterministic_init(load_model(name)(3, 20, Cfg()), 42).to(device).train()
    f = torch.from_numpy(feats).to(device).to(dtype)
    x = ME.SparseTensor(f, torch.from_numpy(coords).to(device))
    logits, fmap = m(x)
    loss = torch.nn.functional.cross_entropy(logits.F.float(), torch.from_numpy(labels).to(device), ignore_index=-1)
    loss.backward()
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters()}
    return logits.F.detach().float().cpu().numpy(), float(loss), grads, m


@pytest.mark.parametrize("name", ["Res16UNet14A", "Res16UNet34C"])
def test_logits_match_reference_fixture(name):
    """HIP engine vs the logits the REFERENCE's model files produced on the oracle backend."""
    fx = np.load(os.path.join(G, "%s_forward.npz" % name.lower()))
    m = deterministic_init(load_model(name)(3, 20, Cfg()), 42).to(DEV).train()
    x = ME.SparseTensor(torch.from_numpy(fx["feats"]).to(DEV), torch.from_numpy(fx["coords"]).to(DEV))
    logits, fmap = m(x)
    err = np.abs(logits.F.detach().cpu().numpy() - fx["logits"]).max()
    assert err < 1e-3, err                          # north_star: logits within 1e-3 fp32
    assert np.abs(m.bn0.bn.running_mean.cpu().numpy() - fx["running_mean_bn0"]).max() < 1e-5


def test_forward_backward_fp32_matches_oracle():
    fx = np.load(os.path.join(G, "res16unet14a_forward.npz"))
    coords, feats = fx["coords"], fx["feats"]
    labels = np.random.default_rng(0).integers(-1, 20, coords.shape[0]).astype(np.int64)
    h_logits, h_loss, h_g, _ = run_model("Res16UNet14A", coords, feats, labels, DEV)
    prev = ME.set_backend(OracleBackend("c"))
    try:
        o_logits, o_loss, o_g, _ = run_model("Res16UNet14A", coords, feats, labels, "cpu")
    finally:
        ME.set_backend(prev)
    assert np.abs(h_logits - o_logits).max() < 1e-3
    assert abs(h_loss - o_loss) < 1e-4
    worst = 0.0
    for k in o_g:
        # relative L2 error per tensor (a single ReLU gate flipping on a 1e-7 difference moves one element by O(1),
        # so max-abs is only held to a looser bound)
        e = np.linalg.norm(h_g[k] - o_g[k]) / max(1e-12, np.linalg.norm(o_g[k]))
        m = np.abs(h_g[k] - o_g[k]).max() / max(1e-6, np.abs(o_g[k]).max())
        worst = max(worst, e)
        assert e < 2e-3 and m < 3e-2, (k, e, m)
    print("worst relative L2 grad error", worst)


def test_bf16_storage_deviation_from_fp32_oracle_is_reported():
    fx = np.load(os.path.join(G, "res16unet14a_forward.npz"))
    coords, feats = fx["coords"], fx["feats"]
    labels = np.random.default_rng(0).integers(-1, 20, coords.shape[0]).astype(np.int64)
    h_logits, h_loss, h_g, _ = run_model("Res16UNet14A", coords, feats, labels, DEV, dtype=torch.bfloat16)
    err = np.abs(h_logits - fx["logits"]).max() / np.abs(fx["logits"]).max()
    print("bf16 storage: max relative logit deviation from the fp32 reference fixture = %.4f" % err)
    assert err < 0.1
    assert np.isfinite(h_loss) and all(np.isfinite(v).all() for v in h_g.values())


def test_reference_style_unfused_calls_equal_fused():
    """reference code calls bn(x); relu(x); out += residual separately -- same numbers as the fused call"""
    torch.manual_seed(0)
    fx = np.load(os.path.join(G, "res16unet14a_forward.npz"))
    x = ME.SparseTensor(torch.randn(fx["coords"].shape[0], 32, device=DEV), torch.from_numpy(fx["coords"]).to(DEV))
    r = ME.SparseTensor(torch.randn(fx["coords"].shape[0], 32, device=DEV), coordinate_map_key=x.coordinate_map_key,
                        coordinate_manager=x.coordinate_manager)
    bn = ME.MinkowskiBatchNorm(32).to(DEV)
    relu = ME.MinkowskiReLU(inplace=True)
    a = bn(x, relu=True, residual=r).F
    out = bn(x)
    out += r
    b = relu(out).F
    assert torch.allclose(a, b, atol=1e-6)


def test_bucket_slot_gradients_on_side_stream_equal_plain_autograd():
    """BucketedDDP (world 1): weight gradients are written straight into the flat bucket on a side stream and BN
    parameter gradients into their slots; they must be bit-identical to the plain autograd path."""
    from languagegroundedsemseg_amd.ddp import BucketedDDP
    fx = np.load(os.path.join(G, "res16unet14a_forward.npz"))
    coords, feats = torch.from_numpy(fx["coords"]).to(DEV), torch.from_numpy(fx["feats"]).to(DEV)
    labels = torch.from_numpy(np.random.default_rng(0).integers(-1, 20, fx["coords"].shape[0]).astype(np.int64)).to(DEV)
    out = []
    for use_ddp in (False, True):
        m = deterministic_init(load_model("Res16UNet14A")(3, 20, Cfg()), 42).to(DEV).train()
        ddp = BucketedDDP(m) if use_ddp else None
        for step in range(2):                      # two steps: slots are reused, .grad reset to None in between
            if ddp is not None:
                ddp.zero_grad()
            else:
                m.zero_grad(set_to_none=True)
            x = ME.SparseTensor(feats.bfloat16(), coords)
            logits, _ = m(x)
            loss = torch.nn.functional.cross_entropy(logits.F.float(), labels, ignore_index=-1)
            loss.backward()
            if ddp is not None:
                ddp.finalize()
        torch.cuda.synchronize()
        out.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
        if ddp is not None:                        # engine-produced gradients really live inside the flat buckets
            inside = 0
            for b in ddp.buckets:
                lo, hi = b["flat"].data_ptr(), b["flat"].data_ptr() + b["flat"].numel() * 4
                for p_, off_ in b["views"]:
                    if p_.grad is not None and lo <= p_.grad.data_ptr() < hi:
                        assert p_.grad.data_ptr() == lo + off_ * 4
                        inside += 1
            assert inside >= 33 + 2 * 32, inside   # every conv kernel and every BN weight/bias of Res16UNet14A
    assert out[0].keys() == out[1].keys()
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


def test_clip_pretrain_step_matches_oracle():
    """BASELINE config[2] shape: representation model (ReLU-free last block, representation_only) + the
    contrastive CLIP loss on the MFMA contraction, forward + backward, HIP engine vs CPU oracle (fp32)."""
    from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
    fx = np.load(os.path.join(G, "res16unet14a_forward.npz"))
    coords, feats = fx["coords"], fx["feats"]
    rng = np.random.default_rng(1)
    labels = rng.integers(-1, 200, coords.shape[0]).astype(np.int64)
    anchors = rng.standard_normal((200, 96)).astype(np.float32)
    crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
    neg = crit.sample_negatives(torch.from_numpy(labels), generator=torch.Generator().manual_seed(5))

    def run(device):
        m = deterministic_init(load_model("Res16UNet34CR")(3, 20, Cfg()), 42).to(device).train()
        m.representation_only(True)
        x = ME.SparseTensor(torch.from_numpy(feats).to(device), torch.from_numpy(coords).to(device))
        out = m(x)
        loss, pos, ngl = crit(out.F.float(), torch.from_numpy(labels).to(device), torch.from_numpy(anchors).to(device),
                              neg_indices=neg.to(device))
        loss.backward()
        return float(loss), out.F.detach().float().cpu().numpy(), m.block8[1].conv2.kernel.grad.detach().cpu().numpy()

    h = run(DEV)
    prev = ME.set_backend(OracleBackend("c"))
    try:
        o = run("cpu")
    finally:
        ME.set_backend(prev)
    assert abs(h[0] - o[0]) < 1e-4
    assert np.abs(h[1] - o[1]).max() < 1e-3
    assert np.linalg.norm(h[2] - o[2]) / np.linalg.norm(o[2]) < 2e-3


def test_bf16_training_step_is_bitwise_reproducible():
    """no atomics on floating-point data anywhere (fixed-order partial sums in wgrad / BN / slot-split conv, weight
    gradients written by exactly one wave): the same step run twice gives bit-identical logits and gradients"""
    from languagegroundedsemseg_amd.synthetic import make_batch
    coords, feats, _ = make_batch([1, 2], voxel=0.02, n_target=60000)
    labels = np.random.default_rng(0).integers(-1, 20, coords.shape[0]).astype(np.int64)
    runs = []
    for _ in range(2):
        logits, loss, grads, _ = run_model("Res16UNet14A", coords, feats, labels, DEV, dtype=torch.bfloat16)
        runs.append((logits, loss, grads))
    assert np.array_equal(runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1]
    for k in runs[0][2]:
        assert np.array_equal(runs[0][2][k], runs[1][2][k]), k


def test_insseg_head_matches_reference_fixture_on_the_engine():
    """downstream/insseg: offsets + logits of the HIP engine vs the fixture the reference's insseg model
    produced on the oracle backend; backward through the 3-channel 1x1 head runs"""
    from languagegroundedsemseg_amd.losses import instance_offset_losses
    fx = np.load(os.path.join(G, "insseg_res16unet14a_forward.npz"))
    m = deterministic_init(load_model("InsSegRes16UNet14A")(3, 20, Cfg()), 42).to(DEV).train()
    x = ME.SparseTensor(torch.from_numpy(fx["feats"]).to(DEV), torch.from_numpy(fx["coords"]).to(DEV))
    off, logits, feats = m(x)
    assert np.abs(off.F.detach().cpu().numpy() - fx["offsets"]).max() < 1e-3
    assert np.abs(logits.F.detach().cpu().numpy() - fx["logits"]).max() < 1e-3
    nl, dl = instance_offset_losses(off.F, torch.from_numpy(fx["coords"][:, 1:]).to(DEV), torch.from_numpy(fx["centers"]).to(DEV),
                                    torch.from_numpy(fx["inst"]).to(DEV), float(fx["voxel"]))
    assert abs(float(nl) - float(fx["norm_loss"])) < 1e-3 and abs(float(dl) - float(fx["dir_loss"])) < 1e-3
    (nl + dl + logits.F.float().square().mean()).backward()
    assert m.offsets.kernel.grad is not None and torch.isfinite(m.offsets.kernel.grad).all()
    assert torch.isfinite(m.conv0p1s1.kernel.grad).all()


def test_fused_flat_sgd_on_device_equals_torch_sgd():
    """FlatSGD's one-kernel bucket update (lgs_sgd_step) == torch.optim.SGD(momentum, dampening, weight_decay) of
    lib/solvers.py over several steps, including a parameter that never receives a gradient"""
    import torch.nn as nn
    from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.unused = nn.Linear(8, 16), nn.Linear(16, 5), nn.Linear(3, 3)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))
    torch.manual_seed(0)
    a, b = Net().to(DEV), Net().to(DEV)
    b.load_state_dict(a.state_dict())
    ddp = BucketedDDP(a, bucket_mb=0.0005)
    fo = FlatSGD(ddp, lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    to = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-2)
    x, y = torch.randn(16, 8, device=DEV), torch.randn(16, 5, device=DEV)
    for step in range(4):
        ddp.zero_grad()
        ((a(x) - y) ** 2).mean().backward()
        ddp.finalize()
        fo.step()
        to.zero_grad(set_to_none=True)
        ((b(x) - y) ** 2).mean().backward()
        to.step()
    for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(p1, p2, atol=1e-6), n1
    assert torch.equal(a.unused.weight, b.unused.weight)
