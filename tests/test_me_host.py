"""Host logic of the MinkowskiEngine-compatible surface (no GPU): data-side utils and SparseTensor
semantics, run on the CPU oracle backend."""
import numpy as np
import pytest
import torch

import MinkowskiEngine as ME
from oracle.backend import OracleBackend


@pytest.fixture()
def oracle_backend():
    prev = ME.set_backend(OracleBackend("c"))
    yield
    ME.set_backend(prev)


def test_sparse_quantize_first_occurrence_ascending():
    pts = np.array([[0.2, 0.1, 0.9], [1.5, 0.0, 0.0], [0.7, 0.9, 0.1], [-0.2, 0.0, 0.0], [1.1, 0.3, 0.9]])
    c, idx = ME.utils.sparse_quantize(pts, return_index=True)
    assert idx.tolist() == [0, 1, 3]
    assert c.tolist() == [[0, 0, 0], [1, 0, 0], [-1, 0, 0]]
    # label collision -> ignore label (lib/voxelizer.py:142 passes ignore_label)
    labels = np.array([3, 4, 5, 6, 4])
    c, l, idx = ME.utils.sparse_quantize(pts, labels=labels, return_index=True, ignore_label=255)
    assert l.tolist() == [255, 4, 6]


def test_sparse_collate_batch_in_column_zero():
    a = torch.tensor([[1, 2, 3], [4, 5, 6]], dtype=torch.int32)
    b = torch.tensor([[7, 8, 9]], dtype=torch.int32)
    bc, f, l = ME.utils.sparse_collate([a, b], [torch.ones(2, 3), torch.zeros(1, 3)], [torch.tensor([1, 2]), torch.tensor([3])])
    assert bc.dtype == torch.int32 and bc[:, 0].tolist() == [0, 0, 1] and bc[2, 1:].tolist() == [7, 8, 9]
    assert f.shape == (3, 3) and l.tolist() == [1, 2, 3]


def test_sparse_tensor_dedup_and_ops(oracle_backend):
    coords = torch.tensor([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1], [1, 1, 1, 1]], dtype=torch.int32)
    feats = torch.arange(8, dtype=torch.float32).reshape(4, 2)
    x = ME.SparseTensor(feats, coords)
    assert x.F.shape == (3, 2) and x.F[:, 0].tolist() == [0., 2., 6.]      # RANDOM_SUBSAMPLE = first occurrence
    assert x.C.tolist() == [[0, 1, 1, 1], [0, 2, 2, 2], [1, 1, 1, 1]]
    assert x.tensor_stride == [1, 1, 1]
    y = ME.SparseTensor(torch.ones(3, 2), coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
    z = ME.cat(x, y)
    assert z.F.shape == (3, 4) and z.coordinate_map_key == x.coordinate_map_key
    y += x
    assert y.F[:, 0].tolist() == [1., 3., 7.]
    other = ME.SparseTensor(torch.ones(3, 2), coords[[0, 1, 3]])
    with pytest.raises(ValueError):
        x + other


def test_conv_stride_and_transpose_keys(oracle_backend):
    rng = np.random.default_rng(0)
    c = np.unique(rng.integers(-6, 6, (200, 3)), axis=0)
    coords = torch.from_numpy(np.concatenate([np.zeros((c.shape[0], 1)), c], 1).astype(np.int32))
    x = ME.SparseTensor(torch.randn(coords.shape[0], 4), coords)
    down = ME.MinkowskiConvolution(4, 8, kernel_size=2, stride=2, dimension=3)
    up = ME.MinkowskiConvolutionTranspose(8, 4, kernel_size=2, stride=2, dimension=3)
    y = down(x)
    assert y.tensor_stride == [2, 2, 2] and (y.C[:, 1:] % 2 == 0).all()
    z = up(y)
    assert z.coordinate_map_key == x.coordinate_map_key            # lands on the cached fine map (needed by me.cat)
    ME.cat(z, x)
    up2 = ME.MinkowskiConvolutionTranspose(4, 4, kernel_size=2, stride=2, dimension=3)
    with pytest.raises(RuntimeError):
        up2(x)                                                       # no finer map cached


def test_region_type_int_constructible():
    assert [ME.RegionType(m) for m in range(3)] == [ME.RegionType.HYPER_CUBE, ME.RegionType.HYPER_CROSS, ME.RegionType.CUSTOM]
