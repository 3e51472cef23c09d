"""Generates tests/golden/trajectory_14a.npz: the fp32 CPU oracle's loss curve for the 30-step training trajectory of
tests/test_gpu_parity_r3.py::test_bf16_storage_trains_like_fp32_over_30_steps (Res16UNet14A, one 5 cm scene of ~12 k voxels,
scene seed 3, structured labels, weights deterministic_init(42), SGD as lib/solvers.py configures it).

Thirty oracle steps take ~5 minutes of CPU; the GPU test re-runs only the first 3 of them live, checks that they reproduce the
recorded values (same scene, same weights, same oracle) and holds the HIP fp32 / bf16 curves to bands around the recorded
curve.

    python tests/golden/make_trajectory.py        (CPU only; needs nothing outside this repository)
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
from oracle.backend import OracleBackend  # noqa: E402
from languagegroundedsemseg_amd.synthetic import make_batch  # noqa: E402
from test_gpu_parity_r2 import structured_labels  # noqa: E402
from test_gpu_parity_r3 import _trajectory, TRAJ_SCENE  # noqa: E402

if __name__ == "__main__":
    coords, feats, _ = make_batch(**TRAJ_SCENE)
    labels = structured_labels(coords)
    ME.set_backend(OracleBackend("torch"))
    t0 = time.time()
    o32 = _trajectory("cpu", torch.float32, coords, feats, labels, 30, False)
    print("oracle fp32 curve (%.0f s):" % (time.time() - t0), o32)
    np.savez(os.path.join(HERE, "trajectory_14a.npz"), oracle_fp32=o32.astype(np.float64), voxels=np.int64(coords.shape[0]))
