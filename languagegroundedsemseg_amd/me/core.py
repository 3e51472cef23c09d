"""SparseTensor / CoordinateManager / CoordinateMapKey -- host-side mirror of the MinkowskiEngine 0.5.4
Python types the reference is written against:
  SparseTensor(features, coordinates)                 lib/train_test/pl_BaselineTrainer.py:300
  SparseTensor(features, coordinate_map_key=, coordinate_manager=)   models/wrapper.py:27
  .F .C .coordinate_map_key .coordinate_manager        pl_BaselineTrainer.py:384
  `out += residual` on the same map                    models/modules/resnet_block.py:54
The arithmetic lives behind a backend (backend_hip.HipBackend = the C-ABI engine).  The default and
only product backend is HIP; tests install the CPU oracle through set_backend() to check it.
"""
from enum import Enum

import torch

from . import deferred as _deferred

_BACKEND = None


def get_backend():
    global _BACKEND
    if _BACKEND is None:
        from .backend_hip import HipBackend
        _BACKEND = HipBackend()
    return _BACKEND


def set_backend(backend):
    """Install a backend object (tests only: the CPU oracle). Returns the previous one."""
    global _BACKEND
    _deferred.flush_all()        # recorded calls run on the backend they were recorded under
    prev, _BACKEND = _BACKEND, backend
    return prev


class SparseTensorQuantizationMode(Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3


class CoordinateMapKey:
    """(tensor_stride, id) handle of one coordinate map inside a manager."""

    def __init__(self, tensor_stride, key_id, D=3):
        self._ts = [int(tensor_stride)] * D if isinstance(tensor_stride, int) else [int(t) for t in tensor_stride]
        self.id = key_id

    def get_tensor_stride(self):
        return list(self._ts)

    def get_key(self):
        return (tuple(self._ts), str(self.id))

    def get_coordinate_size(self):
        return len(self._ts) + 1

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and other.id == self.id and other._ts == self._ts

    def __hash__(self):
        return hash((tuple(self._ts), self.id))

    def __repr__(self):
        return "coordinate map key:%s, id:%s" % (self._ts, self.id)


class CoordinateManager:
    """One per input batch.  All tensors at one resolution share one map, so `me.cat` and `+=` can
    assert identical keys."""

    def __init__(self, D=3, device=None, backend=None):
        self.D = D
        self.backend = backend or get_backend()
        self.device = torch.device(device) if device is not None else None
        self._m = None
        self._keys = {}
        self._pending = []          # recorded, not yet executed operations on tensors of this manager (me/deferred.py)
        self._draining = self._redrain = False

    def _ensure(self, device):
        if self._m is None:
            self.device = torch.device(device)
            self._m = self.backend.new_manager(self.device)
        return self._m

    def _key(self, kid):
        if kid not in self._keys:
            self._keys[kid] = CoordinateMapKey(self._m.tensor_stride(kid), kid, self.D)
        return self._keys[kid]

    # -- ME API
    def insert_and_map(self, coordinates, tensor_stride=1, string_id=""):
        m = self._ensure(coordinates.device)
        kid, nu, unique_index, inverse = m.insert(coordinates)
        return self._key(kid), (unique_index, inverse)

    def stride(self, key, stride=2):
        s = stride if isinstance(stride, int) else int(stride[0])
        assert s == 2, "only stride 2 is part of the model family"
        return self._key(self._m.stride2(key.id))

    def finer_key(self, key):
        fk = self._m.parent_of(key.id)
        if fk < 0:
            raise RuntimeError("transposed convolution needs a cached finer coordinate map; none exists for %r" % key)
        return self._key(fk)

    def size(self, key):
        return self._m.map_size(key.id)

    def get_coordinates(self, key):
        return self._m.coords(key.id)

    def kernel_map_handle(self, in_key, out_key, kernel_size):
        return self._m.kernel_map(in_key.id, out_key.id, kernel_size)

    def kernel_map(self, in_key, out_key, stride=1, kernel_size=3, **kwargs):
        """-> {k: int tensor [2, M_k]} (in_row; out_row), the ME `kernel_map` query."""
        h = self._m.kernel_map(in_key.id, out_key.id, kernel_size)
        k, i, o = h.export()
        out = {}
        for kk in torch.unique(k).tolist():
            sel = k == kk
            out[int(kk)] = torch.stack([i[sel], o[sel]])
        return out


class SparseTensor:
    # deferred execution (me/deferred.py): while `_op` is set the features do not exist yet -- the tensor is the future result of
    # a recorded convolution / norm / concat; reading `.F` (or anything derived from it) executes the manager's queue
    _op = None
    _meta = None           # (channels, dtype, device) of a pending tensor
    _cat_slot = None
    _bn_stats = None

    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_map_key=None, coordinate_manager=None,
                 quantization_mode=SparseTensorQuantizationMode.RANDOM_SUBSAMPLE, allow_duplicate_coordinates=False,
                 minkowski_algorithm=None, requires_grad=None, device=None):
        assert isinstance(features, torch.Tensor) and features.dim() == 2, "features must be a [N, C] tensor"
        if device is not None:
            features = features.to(device)
        self.quantization_mode = quantization_mode
        if coordinates is not None:
            assert coordinate_map_key is None
            if not isinstance(coordinates, torch.Tensor):
                coordinates = torch.as_tensor(coordinates)
            if coordinates.dtype not in (torch.int32,):
                # ME warns and floors when the coordinates are not IntTensor
                coordinates = torch.floor(coordinates.double()).to(torch.int32) if coordinates.is_floating_point() \
                    else coordinates.to(torch.int32)
            coordinates = coordinates.to(features.device)
            assert coordinates.shape[0] == features.shape[0], "coordinates / features row mismatch"
            D = coordinates.shape[1] - 1
            if coordinate_manager is None:
                coordinate_manager = CoordinateManager(D=D, device=features.device)
            key, (unique_index, inverse) = coordinate_manager.insert_and_map(coordinates, tensor_stride)
            self.unique_index, self.inverse_mapping = unique_index, inverse
            if unique_index.shape[0] != features.shape[0]:
                if quantization_mode == SparseTensorQuantizationMode.RANDOM_SUBSAMPLE:
                    features = features[unique_index]
                elif quantization_mode in (SparseTensorQuantizationMode.UNWEIGHTED_SUM,
                                           SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE):
                    acc = torch.zeros((unique_index.shape[0], features.shape[1]), dtype=features.dtype, device=features.device)
                    acc.index_add_(0, inverse, features)
                    if quantization_mode == SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE:
                        cnt = torch.zeros(unique_index.shape[0], dtype=features.dtype, device=features.device)
                        cnt.index_add_(0, inverse, torch.ones_like(inverse, dtype=features.dtype))
                        acc = acc / cnt.unsqueeze(1)
                    features = acc
                else:
                    raise ValueError("duplicate coordinates with NO_QUANTIZATION")
            coordinate_map_key = key
        else:
            assert coordinate_map_key is not None and coordinate_manager is not None, \
                "either coordinates or (coordinate_map_key, coordinate_manager) must be given"
        self._F = features
        if requires_grad is not None:
            self._F.requires_grad_(requires_grad)
        self.coordinate_map_key = coordinate_map_key
        self._manager = coordinate_manager

    @classmethod
    def _pending(cls, coordinate_map_key, coordinate_manager, meta):
        """a tensor whose features are the result of an operation still in the manager's deferred queue"""
        t = object.__new__(cls)
        t._F = None
        t._meta = meta
        t.coordinate_map_key = coordinate_map_key
        t._manager = coordinate_manager
        t.quantization_mode = SparseTensorQuantizationMode.RANDOM_SUBSAMPLE
        return t

    def _nch(self):
        """channel count without forcing a pending tensor"""
        return self._F.shape[1] if self._op is None else self._meta[0]

    # -- ME attribute surface
    @property
    def F(self):
        if self._op is not None:
            _deferred.materialise(self)
        return self._F

    @property
    def feats(self):
        return self.F

    @property
    def features(self):
        return self.F

    @property
    def C(self):
        return self._manager.get_coordinates(self.coordinate_map_key)

    @property
    def coordinates(self):
        return self.C

    @property
    def coords(self):
        return self.C

    @property
    def coordinate_manager(self):
        return self._manager

    @property
    def coords_man(self):
        return self._manager

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def D(self):
        return self._manager.D

    @property
    def dimension(self):
        return self._manager.D

    @property
    def device(self):
        return self.F.device

    @property
    def dtype(self):
        return self.F.dtype

    @property
    def shape(self):
        return self.F.shape

    @property
    def requires_grad(self):
        return self.F.requires_grad

    def size(self, *a):
        return self.F.size(*a)

    def __len__(self):
        return self.F.shape[0]

    def _like(self, feats):
        return SparseTensor(feats, coordinate_map_key=self.coordinate_map_key, coordinate_manager=self._manager)

    def _check(self, other):
        if not (other._manager is self._manager and other.coordinate_map_key == self.coordinate_map_key):
            raise ValueError("SparseTensors must share the coordinate manager and the coordinate map key")

    def _binary(self, other, op):
        if isinstance(other, SparseTensor):
            self._check(other)
            return self._like(op(self.F, other.F))
        return self._like(op(self.F, other))

    def __add__(self, o):
        return self._binary(o, torch.add)

    __radd__ = __add__

    def __sub__(self, o):
        return self._binary(o, torch.sub)

    def __mul__(self, o):
        return self._binary(o, torch.mul)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._binary(o, torch.div)

    def __neg__(self):
        return self._like(-self.F)

    def __iadd__(self, o):
        # `out += residual` (resnet_block.py:54).  On the pending result of a norm the addend becomes that norm's residual
        # operand (one fused kernel, me/deferred.py); otherwise in place on .F -- autograd-legal because the conv / BN
        # functions save their own inputs, never their outputs' storage
        if isinstance(o, SparseTensor):
            self._check(o)
            if self._op is not None and _deferred.add_residual(self, o):
                return self
            o = o.F
        f = self.F
        self._F = f + o if f.requires_grad and f.is_leaf else f.add_(o)
        return self

    def __isub__(self, o):
        if isinstance(o, SparseTensor):
            self._check(o)
            o = o.F
        self._F = self.F.sub_(o)
        return self

    def detach(self):
        return self._like(self.F.detach())

    def to(self, *a, **k):
        return self._like(self.F.to(*a, **k))

    def float(self):
        return self._like(self.F.float())

    def bfloat16(self):
        return self._like(self.F.bfloat16())

    def features_at(self, batch_index):
        c = self.C
        return self.F[c[:, 0] == batch_index]

    def coordinates_at(self, batch_index):
        c = self.C
        return c[c[:, 0] == batch_index][:, 1:]

    @property
    def decomposed_features(self):
        c = self.C
        nb = int(c[:, 0].max().item()) + 1 if c.shape[0] else 0
        return [self.F[c[:, 0] == b] for b in range(nb)]

    @property
    def decomposed_coordinates(self):
        c = self.C
        nb = int(c[:, 0].max().item()) + 1 if c.shape[0] else 0
        return [c[c[:, 0] == b][:, 1:] for b in range(nb)]

    def __repr__(self):
        return "SparseTensor(F=%s, %r, device=%s)" % (tuple(self.F.shape), self.coordinate_map_key, self.F.device)


def cat(*sparse_tensors):
    """Channel concat of tensors living on the same coordinate map, order preserved
    (res16unet.py:237,247,257,267: (upsampled, skip))."""
    if len(sparse_tensors) == 1 and isinstance(sparse_tensors[0], (list, tuple)):
        sparse_tensors = tuple(sparse_tensors[0])
    first = sparse_tensors[0]
    for s in sparse_tensors[1:]:
        first._check(s)
    if _deferred.ENABLED and any(t._op is not None for t in sparse_tensors):
        return _deferred.record_cat(sparse_tensors)
    return cat_now(sparse_tensors)


def cat_now(sparse_tensors, out=None):
    """the concat itself (all inputs materialised); `out`: the pending tensor to fill (me/deferred.py)"""
    first = sparse_tensors[0]
    f = _cat_features(sparse_tensors)
    if out is None:
        return first._like(f)
    out._F, out._op = f, None
    return out


def _cat_features(sparse_tensors):
    if len(sparse_tensors) == 2:
        a, b = (t._cat_slot for t in sparse_tensors)
        if a is not None and a.off == 0 and sparse_tensors[0].F.data_ptr() == a.buf.data_ptr():
            # zero-copy: both halves were written straight into one [N, C1 + C2] buffer by their norms (me.modules) ...
            if (b is not None and a.buf is b.buf and b.off == a.width and a.width + b.width == a.buf.shape[1]):
                return _CatViewFunction.apply(sparse_tensors[0].F, sparse_tensors[1].F, a)
            # ... or the first half was, and the second was copied in beside it when that norm ran
            if a.partner is sparse_tensors[1] and a.width + sparse_tensors[1].F.shape[1] == a.buf.shape[1]:
                return _CatViewFunction.apply(sparse_tensors[0].F, sparse_tensors[1].F, a)
    return torch.cat([s.F for s in sparse_tensors], dim=1)


class _CatViewFunction(torch.autograd.Function):
    """ME.cat of two tensors that already live side by side in one buffer: forward returns the buffer, backward hands
    each input its column slice of the gradient (read in place by the consumers' row-stride support)."""

    @staticmethod
    def forward(ctx, a, b, slot):
        ctx.c1 = a.shape[1]
        return slot.buf.view(slot.buf.shape)      # a fresh tensor object over the same storage (not an input)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c1], g[:, ctx.c1:], None
