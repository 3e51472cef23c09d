"""CPU ORACLE backend for the MinkowskiEngine-compatible surface (TEST INFRASTRUCTURE ONLY).

Implements the same backend protocol as languagegroundedsemseg_amd/me/backend_hip.py on CPU tensors so
that the *same* model code can be run against the oracle:
    prev = ME.set_backend(OracleBackend()); ...; ME.set_backend(prev)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package never does and never selects it on its own.

Two arithmetic flavours for the convolution:
  impl="c"      oracle/sparse_oracle.c (plain loops, double accumulation)          -- the checker
  impl="torch"  per-offset gather -> BLAS GEMM -> scatter-add in fp32 on all host cores, i.e. the algorithm
                of MinkowskiEngine's CPU backend restated with torch ops   -- the timed CPU baseline
Parity status: unpinned against MinkowskiEngine itself (absent); pinned against dense torch conv3d
(tests/test_oracle_dense.py).
"""
import numpy as np
import torch

from . import oracle as orc


class OracleKernelMap:
    def __init__(self, mgr, in_key, out_key, ks, impl):
        self.mgr, self.in_key, self.out_key, self.ks, self.impl = mgr, in_key, out_key, ks, impl
        self.K = ks ** 3
        ci, co = mgr._coords[in_key], mgr._coords[out_key]
        if ks == 1:
            n = ci.shape[0]
            self.km = (np.zeros(n, np.int32), np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64))
        else:
            self.km = orc.kernel_map(ci, co, ks, mgr._ts[in_key])
        self._by_k = None

    def export(self):
        k, i, o = self.km
        return (torch.from_numpy(k.astype(np.int32)), torch.from_numpy(i.astype(np.int32)), torch.from_numpy(o.astype(np.int32)))

    def _dir(self, transposed):
        k, i, o = self.km
        n_in, n_out = self.mgr.map_size(self.in_key), self.mgr.map_size(self.out_key)
        if transposed:
            return (k, o, i), n_out, n_in
        return (k, i, o), n_in, n_out

    def _pairs_by_k(self):
        if self._by_k is None:
            k, i, o = self.km
            self._by_k = []
            for kk in range(self.K):
                sel = np.nonzero(k == kk)[0]
                self._by_k.append((torch.from_numpy(i[sel]), torch.from_numpy(o[sel])))
        return self._by_k

    def _w(self, weight):
        return weight.detach().reshape(self.K, -1, weight.shape[-1]).float()

    def conv_forward(self, x, weight, bias, transposed):
        km, n_in, n_out = self._dir(transposed)
        w = self._w(weight)
        if self.impl == "torch":
            xf = x.float()
            out = torch.zeros((n_out, w.shape[2]), dtype=torch.float32)
            for kk, (pi, po) in enumerate(self._pairs_by_k()):
                if pi.numel() == 0:
                    continue
                src, dst = (po, pi) if transposed else (pi, po)
                out.index_add_(0, dst, xf.index_select(0, src) @ w[kk])
            if bias is not None:
                out += bias.detach().reshape(1, -1).float()
            return out.to(x.dtype)
        b = bias.detach().reshape(-1).float().numpy() if bias is not None else None
        y = orc.conv_forward(x.detach().float().numpy(), w.numpy(), km, n_out, b)
        return torch.from_numpy(y).to(x.dtype)

    def conv_dgrad(self, gout, weight, transposed):
        km, n_in, n_out = self._dir(transposed)
        w = self._w(weight)
        if self.impl == "torch":
            gf = gout.float()
            gin = torch.zeros((n_in, w.shape[1]), dtype=torch.float32)
            for kk, (pi, po) in enumerate(self._pairs_by_k()):
                if pi.numel() == 0:
                    continue
                src, dst = (po, pi) if transposed else (pi, po)
                gin.index_add_(0, src, gf.index_select(0, dst) @ w[kk].t())
            return gin.to(gout.dtype)
        g = orc.conv_dgrad(gout.detach().float().numpy(), w.numpy(), km, n_in)
        return torch.from_numpy(g).to(gout.dtype)

    def conv_wgrad(self, x, gout, transposed):
        km, n_in, n_out = self._dir(transposed)
        if self.impl == "torch":
            xf, gf = x.float(), gout.float()
            gw = torch.zeros((self.K, x.shape[1], gout.shape[1]), dtype=torch.float32)
            for kk, (pi, po) in enumerate(self._pairs_by_k()):
                if pi.numel() == 0:
                    continue
                src, dst = (po, pi) if transposed else (pi, po)
                gw[kk] = xf.index_select(0, src).t() @ gf.index_select(0, dst)
            return gw
        gw = orc.conv_wgrad(x.detach().float().numpy(), gout.detach().float().numpy(), km, self.K)
        return torch.from_numpy(gw)


class OracleManager:
    def __init__(self, device, impl):
        self.device = torch.device("cpu")
        self.impl = impl
        self._coords, self._ts, self._fine, self._coarse, self._kmaps = {}, {}, {}, {}, {}

    def insert(self, coords):
        c = coords.detach().cpu().numpy().astype(np.int32)
        ui, inv = orc.unique_coords(c)
        self._coords[0], self._ts[0], self._fine[0] = np.ascontiguousarray(c[ui]), 1, -1
        return 0, ui.shape[0], torch.from_numpy(ui), torch.from_numpy(inv)

    def stride2(self, key):
        if key in self._coarse:
            return self._coarse[key]
        oc, _ = orc.stride_coords(self._coords[key], self._ts[key] * 2)
        nk = len(self._coords)
        self._coords[nk], self._ts[nk], self._fine[nk] = oc, self._ts[key] * 2, key
        self._coarse[key] = nk
        return nk

    def parent_of(self, key):
        return self._fine[key]

    def map_size(self, key):
        return self._coords[key].shape[0]

    def tensor_stride(self, key):
        return self._ts[key]

    def coords(self, key):
        return torch.from_numpy(self._coords[key])

    def kernel_map(self, in_key, out_key, ks):
        k = (in_key, out_key, ks)
        if k not in self._kmaps:
            self._kmaps[k] = OracleKernelMap(self, in_key, out_key, ks, self.impl)
        return self._kmaps[k]


class OracleBackend:
    """No bn_forward attribute on purpose: MinkowskiBatchNorm then runs its plain nn.BatchNorm1d, which is
    literally what the reference's ME.MinkowskiBatchNorm does."""
    name = "oracle"

    def __init__(self, impl="c"):
        assert impl in ("c", "torch")
        self.impl = impl

    def new_manager(self, device):
        return OracleManager(device, self.impl)

    def clip_similarity(self, feats, anchors):
        f = feats.detach().double()
        a = anchors.detach().double()
        nf = f.norm(dim=1).clamp_min(1e-12)
        s = (f / nf[:, None]) @ (a / a.norm(dim=1, keepdim=True).clamp_min(1e-12)).t()
        return s.float(), (1.0 / nf).float()
