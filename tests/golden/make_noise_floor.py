"""Generates tests/golden/noise_floor_34d_70k.npz: the bf16-STORAGE noise floor of the CPU oracle itself for the
Res16UNet34D + CLIP-loss step on the 70 k-voxel scene of tests/test_gpu_parity_r2.py::test_res16unet34d_clip_step_vs_oracle
(scene seed 9, label seed 2, negative seed 5, weights deterministic_init(42)).

The oracle runs the step twice -- fp32 and with bf16-rounded activations (its own conv / BN / ReLU code, rounding after every
op) -- and the deviation of the second from the first is what bf16 storage costs in this network irrespective of any
kernel: the HIP bf16 step is held to <= 1.25 x that floor.  Two 512-channel oracle passes take several minutes of CPU, so the
GPU test reads these numbers instead of re-running the bf16 pass; it still runs the fp32 pass and checks that ITS loss equals
the one recorded here (same scene, same weights, same oracle).

    python tests/golden/make_noise_floor.py        (CPU only; needs nothing outside this repository)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import MinkowskiEngine as ME  # noqa: E402
from helpers import Cfg, deterministic_init  # noqa: E402
from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss  # noqa: E402
from languagegroundedsemseg_amd.models import load_model  # noqa: E402
from languagegroundedsemseg_amd.synthetic import make_batch, text_anchors  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(1e-30, np.linalg.norm(b.astype(np.float64))))


def clip_step(coords, feats, labels, anchors, neg, dtype):
    crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
    m = deterministic_init(load_model("Res16UNet34D")(3, 20, Cfg()), 42).train()
    m.representation_only(True)
    x = ME.SparseTensor(torch.from_numpy(feats).to(dtype), torch.from_numpy(coords))
    out = m(x)
    loss = crit(out.F, torch.from_numpy(labels), torch.from_numpy(anchors), neg_indices=neg)[0]
    loss.backward()
    grads = {k: p.grad.detach().float().numpy() for k, p in m.named_parameters() if p.grad is not None}
    return float(loss), out.F.detach().float().numpy(), grads


def main():
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    coords, feats, _ = make_batch([9], voxel=0.02, n_target=70000)
    labels = np.random.default_rng(2).integers(-1, 200, coords.shape[0]).astype(np.int64)
    anchors = text_anchors(200, 512)
    neg = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3).sample_negatives(
        torch.from_numpy(labels), generator=torch.Generator().manual_seed(5))
    prev = ME.set_backend(OracleBackend("torch"))
    try:
        t0 = time.time()
        o = clip_step(coords, feats, labels, anchors, neg, torch.float32)
        print("fp32 oracle pass: %.0f s, loss %.6f" % (time.time() - t0, o[0]))
        t0 = time.time()
        b = clip_step(coords, feats, labels, anchors, neg, torch.bfloat16)
        print("bf16-storage oracle pass: %.0f s, loss %.6f" % (time.time() - t0, b[0]))
    finally:
        ME.set_backend(prev)
    names = sorted(o[2])
    per = np.array([rel_l2(b[2][k], o[2][k]) for k in names])
    tot = float(np.sqrt(sum(np.linalg.norm(b[2][k].astype(np.float64) - o[2][k]) ** 2 for k in names) /
                        sum(np.linalg.norm(o[2][k].astype(np.float64)) ** 2 for k in names)))
    feat = rel_l2(b[1], o[1])
    print("bf16-storage oracle vs fp32 oracle: gradient rel-L2 %.4f (median tensor %.4f), feature rel-L2 %.4e" % (tot, float(np.median(per)), feat))
    np.savez_compressed(os.path.join(HERE, "noise_floor_34d_70k.npz"),
                        n_voxels=np.int64(coords.shape[0]), loss_fp32=np.float64(o[0]), loss_bf16=np.float64(b[0]),
                        grad_rel_l2_total=np.float64(tot), grad_rel_l2_per_tensor=per, grad_names=np.array(names),
                        feature_rel_l2=np.float64(feat), grad_norm_fp32=np.array([np.linalg.norm(o[2][k].astype(np.float64)) for k in names]),
                        feature_sample_rows=np.arange(0, coords.shape[0], 997), feature_sample_fp32=o[1][::997].astype(np.float32))


if __name__ == "__main__":
    main()
