"""Device voxelisation against the oracle: lgs_voxelize + engine dedup + lgs_label_vote through
ME.utils.voxelize / ME.utils.sparse_quantize on HIP tensors.  Integer / index work: bit-exact."""
import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from oracle import oracle as orc
from test_quantize_cpu import _points, _rigid

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_voxelize_rigid_transform_bit_exact(seed):
    pts, labels = _points(seed, n=30000)
    M = _rigid(seed + 10, voxel=0.02)
    coords, ui, inv, lab = orc.quantize(pts, M, labels, ignore_label=-1, batch_index=3)
    got = ME.utils.voxelize(torch.from_numpy(pts).to(DEV), affine=M, batch_index=3)
    assert np.array_equal(got.cpu().numpy(), coords)


def test_sparse_quantize_on_device_bit_exact_with_label_collisions():
    pts, labels = _points(7, n=60000)
    q = 0.05
    M = np.eye(4)
    M[:3, :3] /= q
    coords, ui, inv, lab = orc.quantize(pts, M, labels, ignore_label=-1)
    feats = torch.from_numpy(pts).to(DEV)
    c, f, l, idx, inverse = ME.utils.sparse_quantize(torch.from_numpy(pts).to(DEV), feats, torch.from_numpy(labels).to(DEV),
                                                     ignore_label=-1, return_index=True, return_inverse=True, quantization_size=q)
    assert (lab == -1).sum() > 100                       # the case is not vacuous
    assert np.array_equal(idx.cpu().numpy(), ui) and np.array_equal(inverse.cpu().numpy(), inv)
    assert np.array_equal(c.cpu().numpy(), coords[ui][:, 1:])
    assert np.array_equal(l.cpu().numpy(), lab)
    assert torch.equal(f, feats[idx])
    only = ME.utils.sparse_quantize(torch.from_numpy(pts).to(DEV), quantization_size=q, return_maps_only=True)
    assert np.array_equal(only.cpu().numpy(), ui)


def test_device_quantize_edge_cases():
    e = ME.utils.voxelize(torch.zeros(0, 3, device=DEV), quantization_size=0.02)
    assert e.shape == (0, 4)
    one = ME.utils.sparse_quantize(torch.tensor([[0.011, -0.011, 0.0]], device=DEV), quantization_size=0.02)
    assert one.cpu().tolist() == [[0, -1, 0]]
    neg = ME.utils.voxelize(torch.tensor([[-1e-7, -0.02, -0.020001]], device=DEV), quantization_size=0.02)
    ref, _, _, _ = orc.quantize(np.array([[-1e-7, -0.02, -0.020001]], np.float32), np.diag([50.0, 50.0, 50.0, 1.0]))
    assert np.array_equal(neg.cpu().numpy(), ref)
    with pytest.raises(RuntimeError):
        ME.utils.voxelize(torch.zeros(4, 3), quantization_size=0.02)           # host tensors: no silent CPU fallback
