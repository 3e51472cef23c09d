"""HIP backend: the product path.  Every call goes through the C-ABI (include/lgs_engine.h).

Tensors stay owned by torch; the engine borrows raw device pointers for the duration of a call and
enqueues on torch's current HIP stream.  There is no fallback: a CPU tensor or a missing library
raises RuntimeError.
"""
import ctypes
import os
import threading
import weakref

import torch

from .. import engine
from .. import tuning as _tuning

_vp = ctypes.c_void_p


def _stream():
    """raw handle of torch's current HIP stream on the current device (one C call: torch.cuda.current_stream() builds a
    Stream object and resolves the device twice, ~14 us of host time per engine call)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOGUARD = _NoGuard()


def _dev(device):
    """device guard that only switches when the tensor's device is not the current one (the usual case: one process per GPU)"""
    if device.index is None or device.index == torch._C._cuda_getDevice():
        return _NOGUARD
    return torch.cuda.device(device)


def _ptr(t):
    """device address as a plain int (None = NULL): every engine entry point has its argtypes declared (engine.py, checked against
    the header by tests/test_abi.py), so ctypes converts -- building a c_void_p object per argument cost ~1 ms of host time per
    step (2500 pointer arguments)"""
    return t.data_ptr() if t is not None else None


def _dtype_code(t):
    if t.dtype == torch.float32:
        return engine.LGS_F32
    if t.dtype == torch.bfloat16:
        return engine.LGS_BF16
    raise RuntimeError("lgs_engine supports float32 and bfloat16 features, got %s" % t.dtype)


def _require_dev(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must live on a HIP device (got %s): the MI355X engine has no CPU fallback"
                           % (what, t.device))


_PACKED = None


def get_packed():
    global _PACKED
    if _PACKED is None:
        _PACKED = PackedWeights()
    return _PACKED


def _row_strided(t, c):
    """row stride (elements) if `t` [n, c] is a column slice of a wider row-major tensor the kernels can read in place
    (16-byte aligned rows), else None"""
    if t.dim() == 2 and t.shape[1] == c and t.stride(1) == 1 and t.stride(0) > c:
        e = t.element_size()
        if (t.stride(0) * e) % 16 == 0 and t.data_ptr() % 16 == 0:
            return t.stride(0)
    return None


_WS = {}
_BN_WSB = {}


def _bn_ws_bytes(L, n, c):
    b = _BN_WSB.get((n, c))
    if b is None:
        if len(_BN_WSB) > 4096:
            _BN_WSB.clear()
        b = _BN_WSB[(n, c)] = L.lgs_bn_workspace_bytes(n, c)
    return b
# input voxels per batch below which weight gradients are not moved to the side stream (LGS_WGRAD_INLINE_BELOW: tuning knob;
# one 145 k-voxel scene per step: 11.2 -> 10.4 ms, the 1.2 M-voxel batch keeps the side stream: 29.9 vs 30.9 ms)
_WGRAD_INLINE_BELOW = _tuning.host("WGRAD_INLINE_BELOW")


def _ws(nbytes, device, stream=None):
    """Scratch of one engine call: ONE grow-only buffer per (device, current stream).  Every use is confined to the launches
    of a single call, and calls on one stream run in order, so the next call may overwrite it; weight gradients on the side
    stream get their own.  (A fresh torch.empty per call cost ~7 us of allocator time, ~250 times per step.)"""
    if stream is None:
        raw_stream = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch._C._cuda_getDevice())
    else:
        raw_stream = stream.cuda_stream
    key = (device.index, raw_stream)
    t = _WS.get(key)
    nbytes = int(nbytes)
    if t is None or t.numel() < nbytes:
        size = max(nbytes + nbytes // 4, 1 << 20)
        if stream is None:
            t = torch.empty(size, dtype=torch.uint8, device=device)
        else:
            # the buffer must belong to the allocator pool of the stream that uses it: when it is replaced by a larger one, the
            # old block may only be handed out again in that stream's order
            with torch.cuda.stream(stream):
                t = torch.empty(size, dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def _param_event(param):
    """the reusable event of a kernel parameter's weight gradient (me/modules.py conv_weight_grad; BucketedDDP waits on it)"""
    ev = getattr(param, "_lgs_wgrad_event", None)
    if ev is None:
        ev = param._lgs_wgrad_event = torch.cuda.Event()
    return ev


def _raw_event(ev, stream):
    """hipEvent_t of a torch event for the engine to record; torch creates the handle at the first record"""
    h = ev.cuda_event
    if not h:
        ev.record(stream)
        h = ev.cuda_event
    return int(h.value if hasattr(h, "value") else h)


_DESC_FIELDS = [f for f, _ in engine.PackDesc._fields_]


class _PackEntry:
    """one packed weight image: the device buffer, its layout as plain ints (no ctypes pointers: conv modules that own
    entries stay picklable / deep-copyable) and a WEAK reference to the parameter it was packed from"""
    __slots__ = ("buf", "desc", "valid", "param_ref", "__weakref__")

    def __init__(self, buf, desc, param):
        self.buf, self.desc, self.valid, self.param_ref = buf, desc, None, weakref.ref(param)

    def cdesc(self):
        d = engine.PackDesc()
        for f in _DESC_FIELDS:
            setattr(d, f, self.desc[f])
        return d


class PackedWeights:
    """Registry of packed weight images (lgs_conv_pack_desc / lgs_pack_weights_batch).  Every conv module OWNS the
    images of its launch shapes (`_pack_cache` dict on the module, one entry per (direction, layout)); this registry
    only holds them weakly, so a model that is dropped frees its images and is no longer re-packed.  An image is valid
    while (epoch, parameter version, parameter address) are unchanged.  optimiser.step() of FlatSGD -- which updates the
    parameters through the C-ABI, invisible to torch's version counters -- calls repack_all(): ONE launch re-packs every
    live image from the updated weights, so the ~125 per-call pack launches of a step disappear.
    Writes torch's version counter does not see (`p.data.copy_()`, `p.data.uniform_()`, `dist.broadcast(p.data)`, a
    raw-pointer write through the C-ABI) need invalidate(): reset_parameters(), load_state_dict() and BucketedDDP's
    initial broadcast call it; any other `.data` write must be followed by ME.invalidate_packed_weights()."""

    def __init__(self):
        self.entries = weakref.WeakSet()
        self.epoch = 0
        self._table = None          # (device copy of the descriptors, entry list, max_total)
        self._dirty = True
        self.enabled = _tuning.host("PACK_CACHE") != 0

    def invalidate(self):
        """every cached image is re-packed from its parameter on next use"""
        self.epoch += 1

    def lookup(self, cache, km, op, transposed, weight, w32, cin, cout, dt):
        """-> (packed buffer or None, pack_mode)"""
        if not self.enabled or cache is None:
            return None, 0
        # the layout of a launch shape on this map: asked once per map (layers share maps) and per state of the tuning table -- the
        # layout depends on knobs (FP32_SPLIT: 6 instead of 4 bytes per element), and a descriptor cached across engine.tuning(...)
        # would hand a buffer of the old size / a "valid" image of the old layout to the new kernel (advisor, round 5)
        ck = (op, transposed, cin, cout, dt, engine.TUNING_EPOCH)
        d = km._pdesc.get(ck)
        if d is None:
            d = engine.PackDesc()
            engine.check(engine.lib().lgs_conv_pack_desc(km.h, int(op), int(transposed), int(cin), int(cout), int(dt), ctypes.byref(d)))
            km._pdesc[ck] = d
        if d.bytes == 0:
            return None, 0
        key = (op, int(transposed), dt, d.ncp, d.nbp, d.K, cin, cout, int(d.dtype), int(d.bytes))   # (d.dtype: the image's own layout code)
        ent = cache.get(key)
        if ent is None:
            buf = torch.empty(int(d.bytes), dtype=torch.uint8, device=w32.device)
            desc = {f: getattr(d, f) for f in _DESC_FIELDS}
            desc["packed"], desc["weight"] = buf.data_ptr(), w32.data_ptr()
            ent = _PackEntry(buf, desc, weight)
            cache[key] = ent
            self.entries.add(ent)
            self._dirty = True
        stamp = (self.epoch, weight._version, w32.data_ptr())
        if ent.desc["weight"] != w32.data_ptr():
            ent.desc["weight"] = w32.data_ptr()
            self._dirty = True
        if ent.valid == stamp:
            return ent.buf, 2
        ent.valid = stamp
        return ent.buf, 1

    def repack_all(self):
        """after the parameters changed behind torch's back (FlatSGD): new epoch, one batched re-pack of the live images"""
        self.epoch += 1
        if not self.enabled:
            return
        live = []
        for e in list(self.entries):
            p = e.param_ref()
            if p is not None and p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda:
                live.append((e, p))
        if not live:
            self._table = None
            return
        ids = tuple(id(e) for e, _ in live)
        if self._dirty or self._table is None or self._table[1] != ids or any(e.desc["weight"] != p.data_ptr() for e, p in live):
            for e, p in live:
                e.desc["weight"] = p.data_ptr()
            arr = (engine.PackDesc * len(live))(*[e.cdesc() for e, _ in live])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self._table = (host.to(live[0][0].buf.device), ids, max(int(e.desc["total"]) for e, _ in live))
            self._dirty = False
        tab, _, max_total = self._table
        with _dev(tab.device):
            engine.check(engine.lib().lgs_pack_weights_batch(_ptr(tab), len(live), max_total, _stream()))
        for e, p in live:
            e.valid = (self.epoch, p._version, p.data_ptr())


def invalidate_packed_weights():
    """call after writing conv weights behind torch's version counters (`.data` writes, raw-pointer updates).  Recorded ME calls
    that have not run yet (me/deferred.py) read the weights when they execute: they run first, with the weights of their call."""
    from . import deferred as _deferred
    _deferred.flush_all()
    if _PACKED is not None:
        _PACKED.invalidate()


class HipKernelMap:
    def __init__(self, mgr, handle, in_key, out_key, ks):
        self.mgr, self.h, self.in_key, self.out_key, self.ks = mgr, handle, in_key, out_key, ks
        self.K = ks ** 3
        self._wsb = {}            # lgs_conv_workspace_bytes per (cin, cout, dtype, op): several layers share a map
        self._pdesc = {}          # lgs_conv_pack_desc per (op, transposed, cin, cout, dtype)

    def _ws_bytes(self, L, cin, cout, dt, op):
        key = (cin, cout, dt, op)
        b = self._wsb.get(key)
        if b is None:
            b = self._wsb[key] = L.lgs_conv_workspace_bytes(self.h, cin, cout, dt, op)
        return b

    def export(self):
        L = engine.lib()
        m = ctypes.c_int64(0)
        with _dev(self.mgr.device):
            engine.check(L.lgs_kmap_export(self.h, None, None, None, _stream(), ctypes.byref(m)))
            n = m.value
            k = torch.empty(n, dtype=torch.int32, device=self.mgr.device)
            i = torch.empty_like(k)
            o = torch.empty_like(k)
            if n:
                engine.check(L.lgs_kmap_export(self.h, _ptr(k), _ptr(i), _ptr(o), _stream(), ctypes.byref(m)))
        return k, i, o

    def _rows(self, transposed):
        n_in = self.mgr.map_size(self.in_key)
        n_out = self.mgr.map_size(self.out_key)
        return (n_out, n_in) if transposed else (n_in, n_out)

    def _w32(self, weight):
        """-> (fp32 contiguous weight tensor whose address the engine reads, cin, cout); the usual case (an fp32 contiguous
        parameter) is the parameter itself: only its address and shape are used, no view objects are built"""
        cout = weight.shape[-1]
        if weight.dtype == torch.float32 and weight.is_contiguous():
            return weight, weight.numel() // (self.K * cout), cout
        w = weight.detach().reshape(self.K, -1, cout).contiguous().float()
        return w, w.shape[1], cout

    def conv_forward(self, x, weight, bias, transposed, bn_pivot=None, want_bn_stats=False, pack_cache=None):
        """want_bn_stats: also return the per-tile BatchNorm statistics of the output (None if this launch shape cannot
        produce them) -> (out, (partials [rows, 2, cout], pivot) | None)"""
        _require_dev(x, "features")
        L = engine.lib()
        w, cin, cout = self._w32(weight)
        # the skip half of a zero-copy ME.cat is a column slice of the concat buffer: gathered in place through a row stride
        ld = _row_strided(x, cin) if (cin % (8 if x.dtype == torch.bfloat16 else 4) == 0 and cout % 4 == 0) else None
        if ld is None:
            x = x.contiguous()
        n_in, n_out = self._rows(transposed)
        assert x.shape[0] == n_in and x.shape[1] == cin, (x.shape, n_in, cin)
        dt = _dtype_code(x)
        part = None
        with _dev(x.device):
            out = torch.empty((n_out, cout), dtype=x.dtype, device=x.device)
            ws = _ws(self._ws_bytes(L, cin, cout, dt, 0), x.device)
            b = bias.detach().reshape(-1).contiguous().float() if bias is not None else None
            if want_bn_stats:
                rows = L.lgs_conv_bn_partial_rows(self.h, int(transposed), cout, dt)
                if rows > 0:
                    part = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
            piv = bn_pivot if part is not None else None
            pk, mode = get_packed().lookup(pack_cache, self, 0, transposed, weight, w, cin, cout, dt)
            engine.check(L.lgs_conv_forward(self.h, int(transposed), _ptr(x), cin, _ptr(w), cout, _ptr(b), _ptr(out), dt,
                                            _ptr(ws), _ptr(part), _ptr(piv), _ptr(pk), int(mode), int(ld or 0), _stream()))
        if want_bn_stats:
            return out, ((part, piv) if part is not None else None)
        return out

    def conv_dgrad(self, gout, weight, transposed, pack_cache=None, accumulate_into=None):
        """accumulate_into: a caller-owned contiguous [n_in, cin] tensor t (the gradient of a residual branch); the result is
        t += dgrad, rounded like "dgrad, then add", formed in the kernel epilogue where the launch shape has one
        (lgs_conv_dgrad_accumulate; t itself is returned) and by an add into the fresh dgrad tensor otherwise."""
        L = engine.lib()
        gout = gout.contiguous()
        w, cin, cout = self._w32(weight)
        n_in, n_out = self._rows(transposed)
        assert gout.shape[0] == n_out and gout.shape[1] == cout
        dt = _dtype_code(gout)
        acc = accumulate_into
        fuse = (acc is not None and acc.is_contiguous() and acc.dtype == gout.dtype and tuple(acc.shape) == (n_in, cin)
                and L.lgs_conv_dgrad_can_accumulate(self.h, int(transposed), cin, cout, dt))
        with _dev(gout.device):
            gin = acc if fuse else torch.empty((n_in, cin), dtype=gout.dtype, device=gout.device)
            ws = _ws(self._ws_bytes(L, cin, cout, dt, 1), gout.device)
            pk, mode = get_packed().lookup(pack_cache, self, 1, transposed, weight, w, cin, cout, dt)
            fn = L.lgs_conv_dgrad_accumulate if fuse else L.lgs_conv_dgrad
            engine.check(fn(self.h, int(transposed), _ptr(gout), cout, _ptr(w), cin, _ptr(gin), dt, _ptr(ws), _ptr(pk), int(mode), _stream()))
            if acc is not None and not fuse:
                gin += acc
        return gin

    def conv_wgrad(self, x, gout, transposed, out=None, stream=None):
        """-> grad_weight [K,cin,cout] fp32; written straight into `out` (e.g. a view of a gradient bucket) if given.
        stream: enqueue on this torch stream instead of the current one (the caller has ordered it after the producers of
        x / gout; no stream context switch on the host)"""
        L = engine.lib()
        if stream is not None and (not gout.is_contiguous() or (not x.is_contiguous() and _row_strided(x, x.shape[1]) is None)):
            with torch.cuda.stream(stream):         # rare: a copy is needed, make it on the target stream
                return self.conv_wgrad(x, gout, transposed, out=out)
        gout = gout.contiguous()
        cin, cout = x.shape[1], gout.shape[1]
        # strided input (see conv_forward): only the position-stationary bf16 kernel reads it in place
        ld = _row_strided(x, cin) if (x.dtype == torch.bfloat16 and self.ks in (2, 3) and cin % 8 == 0 and cout % 8 == 0) else None
        dt = _dtype_code(x)
        if ld is not None and not L.lgs_conv_wgrad_supports_stride(self.h, int(transposed), cin, cout, dt, int(ld)):
            ld = None      # the position-stationary kernel declines this shape: hand the pair-list kernel a contiguous copy
        if ld is None and not x.is_contiguous():
            if stream is not None:
                with torch.cuda.stream(stream):
                    return self.conv_wgrad(x, gout, transposed, out=out)
            x = x.contiguous()
        assert gout.dtype == x.dtype
        raw = stream.cuda_stream if stream is not None else _stream()
        with _dev(x.device):
            if out is not None:
                assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == self.K * cin * cout
                gw = out
            else:
                gw = torch.empty((self.K, cin, cout), dtype=torch.float32, device=x.device)
            if out is None and stream is not None:
                gw.record_stream(stream)
            ws = _ws(self._ws_bytes(L, cin, cout, dt, 2), x.device, stream)
            engine.check(L.lgs_conv_wgrad(self.h, int(transposed), _ptr(x), cin, _ptr(gout), cout, _ptr(gw), dt, _ptr(ws),
                                          int(ld or 0), raw))
        return gw


class HipManager:
    """One per input batch (ME semantics); owns the coordinate maps and kernel maps on the device."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the MI355X engine needs a HIP device, got %s (no CPU fallback)" % device)
        self.device = device if device.index is not None else torch.device("cuda", torch.cuda.current_device())
        h = _vp(None)
        engine.check(engine.lib().lgs_manager_create(self.device.index, ctypes.byref(h)))
        self.h = h
        self._pid = os.getpid()
        self._sizes = {}
        # kernel-map handles are held WEAKLY here (they hold the manager strongly: autograd contexts keep a map -- and
        # through it the manager -- alive until backward has run): no reference cycle, so the manager and its device
        # memory are released by reference counting the moment the step's tensors and graph are gone, with or without
        # the cyclic garbage collector (bench.py disables it inside the timed region)
        self._kmaps = weakref.WeakValueDictionary()
        self._coords = {}

    def __del__(self):
        try:
            # never touch HIP from a forked child (multiprocessing helpers inherit live Python objects: destroying the
            # parent's manager there segfaults)
            if getattr(self, "h", None) and getattr(self, "_pid", None) == os.getpid():
                engine.lib().lgs_manager_destroy(self.h)
            self.h = None
        except Exception:
            pass

    def check(self):
        """device-side consistency flags of this manager's maps (one synchronisation; tests): 0 = consistent"""
        f = ctypes.c_int(0)
        engine.check(engine.lib().lgs_manager_check(self.h, ctypes.byref(f)))
        return f.value

    def insert(self, coords):
        _require_dev(coords, "coordinates")
        L = engine.lib()
        coords = coords.to(torch.int32).contiguous()
        n = coords.shape[0]
        key, nu = ctypes.c_int(0), ctypes.c_int64(0)
        with _dev(self.device):
            ui = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
            inv = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
            engine.check(L.lgs_manager_insert(self.h, _ptr(coords), n, _ptr(ui), _ptr(inv), _stream(), ctypes.byref(key),
                                              ctypes.byref(nu)))
        self._sizes[key.value] = (nu.value, 1)
        # a batch this small is bound by host work (~250 engine calls per step), not by the GPU: its weight gradients go onto the
        # compute stream (no fork / join events, no second stream to feed) -- see me.modules.conv_weight_grad
        self.inline_wgrad = nu.value < _WGRAD_INLINE_BELOW
        return key.value, nu.value, ui[:nu.value], inv[:n]

    def stride2(self, key):
        L = engine.lib()
        ok, n = ctypes.c_int(0), ctypes.c_int64(0)
        with _dev(self.device):
            engine.check(L.lgs_manager_stride2(self.h, key, _stream(), ctypes.byref(ok), ctypes.byref(n)))
        self._sizes[ok.value] = (n.value, self._sizes[key][1] * 2)
        return ok.value

    def parent_of(self, key):
        fk = ctypes.c_int(-1)
        engine.check(engine.lib().lgs_manager_parent_of(self.h, key, ctypes.byref(fk)))
        return fk.value

    def map_size(self, key):
        return self._sizes[key][0]

    def tensor_stride(self, key):
        return self._sizes[key][1]

    def coords(self, key):
        if key not in self._coords:
            n = self.map_size(key)
            with _dev(self.device):
                c = torch.empty((n, 4), dtype=torch.int32, device=self.device)
                engine.check(engine.lib().lgs_manager_get_coords(self.h, key, _ptr(c), _stream()))
            self._coords[key] = c
        return self._coords[key]

    def kernel_map(self, in_key, out_key, ks):
        k = (in_key, out_key, ks)
        km = self._kmaps.get(k)
        if km is None:
            h = _vp(None)     # the engine caches the map itself: a second request returns the same handle at no cost
            with _dev(self.device):
                engine.check(engine.lib().lgs_manager_kernel_map(self.h, in_key, out_key, ks, _stream(), ctypes.byref(h)))
            km = HipKernelMap(self, h, in_key, out_key, ks)
            self._kmaps[k] = km
        return km


def _written_by_engine(*tensors):
    """The engine updated these tensors through raw pointers (running statistics, num_batches_tracked): bump their autograd
    version counters so that everything keyed on `_version` -- MinkowskiBatchNorm's cached eval-mode [mean | invstd] vector,
    torch's own saved-tensor checks -- sees the write (host-only, no launch)."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


class HipBackend:
    name = "hip"
    bn_counts_batches = True    # lgs_bn_forward increments num_batches_tracked itself
    # lgs_bn_forward can write its output into a column slice of a wider buffer (zero-copy ME.cat; host knob ZERO_COPY_CAT=0: A/B)
    bn_out_into = _tuning.host("ZERO_COPY_CAT") != 0
    # lgs_conv_forward can emit the following BatchNorm's statistics from its epilogue -- measured SLOWER in the step (31.5 vs 30.9 ms: the epilogue work on every conv costs more than the skipped column reduction saves), so off unless LGS_CONV_BN_STATS=1
    conv_bn_stats = _tuning.host("CONV_BN_STATS") in ("1", "big")
    conv_bn_stats_min_bytes = (24 << 20) if _tuning.host("CONV_BN_STATS") == "big" else 0   # experiment: large outputs only

    def want_conv_bn_stats(self, n_rows, cout, esize):
        return self.conv_bn_stats and n_rows * cout * esize >= self.conv_bn_stats_min_bytes

    def __init__(self):
        self._side = {}
        self._cur = {}
        self._fork = {}

    def new_manager(self, device):
        return HipManager(device)

    def invalidate_packed_weights(self):
        """conv weights were written behind torch's version counters (`.data` writes): re-pack on next use"""
        invalidate_packed_weights()

    def weights_updated(self):
        """the optimiser changed the parameters outside autograd: re-pack every cached weight image in one launch"""
        get_packed().repack_all()

    def current_stream(self, device):
        """torch.cuda.current_stream(device) without its per-call Python cost: Stream objects are cached per raw handle"""
        idx = device.index if device.index is not None else torch._C._cuda_getDevice()
        raw = torch._C._cuda_getCurrentRawStream(idx)
        s = self._cur.get((idx, raw))
        if s is None:
            s = self._cur[(idx, raw)] = torch.cuda.current_stream(device)
        return s

    def fork_event(self, device):
        """one reusable event per device for ordering the side stream after the compute stream (record, then wait_event)"""
        idx = device.index if device.index is not None else torch._C._cuda_getDevice()
        ev = self._fork.get(idx)
        if ev is None:
            ev = self._fork[idx] = torch.cuda.Event()
        return ev

    def side_stream(self, device):
        """second HIP stream per device: weight gradients run here, concurrently with the dgrad / BN chain (CU-mask
        partitions for it were measured in round 3: 37.8-57.9 vs 29.8 ms per step, removed)"""
        key = device.index if isinstance(device, torch.device) else torch.device(device).index
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    # ---- fused BN(+residual)(+ReLU): lgs_bn_forward / lgs_bn_backward
    def bn_forward(self, x, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, num_batches_tracked=None,
                   conv_stats=None, out_into=None):
        """conv_stats = (partials, pivot) from conv_forward(want_bn_stats=True): the statistics pass over x is skipped.
        out_into = (buffer [n, C_total], column offset): y is written straight into that column slice (zero-copy ME.cat)
        and returned as a strided view of the buffer."""
        _require_dev(x, "features")
        L = engine.lib()
        x = x.contiguous()
        n, c = x.shape
        if n == 0:      # what nn.BatchNorm1d (= MinkowskiBatchNorm's `.bn`) says about batch statistics of nothing
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (list(x.shape),))
        dt = _dtype_code(x)
        part, piv = conv_stats if conv_stats is not None else (None, None)
        y_ld = 0
        with _dev(x.device):
            if out_into is not None:
                buf, off = out_into
                y = buf[:, off:off + c]
                y_ld = buf.stride(0)
                assert buf.dtype == x.dtype and buf.shape[0] == n and (y_ld * x.element_size()) % 16 == 0 and y.data_ptr() % 16 == 0
            else:
                y = torch.empty_like(x)
            stats = torch.empty(2 * c, dtype=torch.float32, device=x.device)
            res = residual.contiguous() if residual is not None else None
            ws = _ws(_bn_ws_bytes(L, n, c), x.device)
            engine.check(L.lgs_bn_forward(_ptr(x), n, c, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                                          _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _ptr(res), int(relu), _ptr(y),
                                          _ptr(stats), dt, _ptr(ws), _ptr(part), int(part.shape[0]) if part is not None else 0,
                                          _ptr(piv), int(y_ld), _stream()))
        _written_by_engine(running_mean, running_var, num_batches_tracked)
        return y, stats

    def bn_backward(self, x, y, dy, gamma, beta, stats, relu, want_residual, dgamma_out=None, dbeta_out=None):
        """relu: 0 none, 1 mask from y, 2 mask recomputed from x (y may be None)"""
        L = engine.lib()
        n, c = x.shape
        dt = _dtype_code(x)
        # a column slice of a wider row-major tensor (the gradient of one ME.cat input) is read in place
        esz = dy.element_size()
        if (dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) >= c and (dy.stride(0) * esz) % 16 == 0
                and dy.data_ptr() % 16 == 0 and dy.dtype == x.dtype):
            dy_ld = dy.stride(0)
        else:
            dy = dy.contiguous()
            dy_ld = c
        y_ld = 0
        if y is not None:
            y_ld = _row_strided(y, c) or 0
            if y_ld == 0:
                y = y.contiguous()
        with _dev(x.device):
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_residual else None
            dgamma = dgamma_out if dgamma_out is not None else torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = dbeta_out if dbeta_out is not None else torch.empty(c, dtype=torch.float32, device=x.device)
            ws = _ws(_bn_ws_bytes(L, n, c), x.device)
            engine.check(L.lgs_bn_backward(_ptr(x), _ptr(y), _ptr(dy), int(dy_ld), n, c, _ptr(gamma), _ptr(beta), _ptr(stats), int(relu), _ptr(dx),
                                           _ptr(dres), _ptr(dgamma), _ptr(dbeta), dt, _ptr(ws), int(y_ld), _stream()))
        return dx, dres, dgamma, dbeta

    # ---- a whole BasicBlock per call (csrc/lgs_block.hip): small batches, everything on the compute stream
    # host staging of lgs_block_fwd / lgs_block_bwd: one buffer per direction and THREAD (ctypes releases the GIL during the call,
    # so a second Python / autograd thread -- multi-device backward, two models driven from two threads -- must not be able to
    # overwrite the struct the engine is still reading; advisor, round 4)
    _blk_tls = threading.local()

    @classmethod
    def _blk_stage(cls, which):
        """-> (buffer, address) of this thread's staging struct for direction `which` ("f" / "b")"""
        st = getattr(cls._blk_tls, which, None)
        if st is None:
            buf = ctypes.create_string_buffer(512)
            st = (buf, ctypes.addressof(buf))
            setattr(cls._blk_tls, which, st)
        return st

    def _block_ws(self, L, kmap3, kmap1, cin, planes, dt, n, device):
        key = ("cblk_ws", cin, planes, dt)
        b = kmap3._wsb.get(key)
        if b is None:
            b = kmap3._wsb[key] = L.lgs_block_workspace_bytes(kmap3.h, kmap1.h if kmap1 is not None else None, cin, planes, dt)
        # conv scratch and BatchNorm scratch are never live at the same time inside the sequence (as in the call-by-call path,
        # where both are the stream's one grow-only buffer): one buffer, sized for the larger
        return _ws(max(b, _bn_ws_bytes(L, n, planes)), device)

    def block_forward(self, x, kmap3, kmap1, ws3, pcs, norms, affines, relu_final):
        """BasicBlock forward through lgs_block_forward -> (o1, st1, y1, o2, st2, y2, od, std).  The argument struct is packed
        with ONE struct.pack_into call (setting ~50 ctypes fields one by one cost 60 us per block, more than the calls it saves)"""
        L = engine.lib()
        self.block_calls = getattr(self, "block_calls", 0) + 1
        w1, w2, wd = ws3
        n, cin = x.shape
        planes = w1.shape[2]
        dt = _dtype_code(x)
        dev = x.device
        ds = wd is not None
        pk = get_packed()
        with _dev(dev):
            buf = torch.empty((6 if ds else 4, n, planes), dtype=x.dtype, device=dev)
            st = torch.empty((3 if ds else 2, 2 * planes), dtype=torch.float32, device=dev)
            p1, pm1 = pk.lookup(pcs[0], kmap3, 0, False, w1, w1, cin, planes, dt)
            p2, pm2 = pk.lookup(pcs[1], kmap3, 0, False, w2, w2, planes, planes, dt)
            pd, pmd = pk.lookup(pcs[2], kmap1, 0, False, wd, wd, cin, planes, dt) if ds else (None, 0)
            b0 = buf.data_ptr()
            row = n * planes * x.element_size()
            s0 = st.data_ptr()
            srow = 2 * planes * 4
            bn = []
            touched = []
            for m, (g, b) in zip(norms, affines):
                if m is None:
                    bn += [0, 0, 0, 0, 0, 0.0, 0.0]
                    continue
                bn += [g.data_ptr(), b.data_ptr(), _ptr(m.running_mean) or 0, _ptr(m.running_var) or 0, _ptr(m.num_batches_tracked) or 0,
                       float(m.eps), float(m.momentum)]
                touched += [m.running_mean, m.running_var, m.num_batches_tracked]
            cws = self._block_ws(L, kmap3, kmap1, cin, planes, dt, n, dev).data_ptr()
            args, addr = self._blk_stage("f")
            engine.BLOCK_FWD_PACK.pack_into(
                args, 0, kmap3.h.value, kmap1.h.value if ds else 0, dt, int(relu_final), cin, planes, n, x.data_ptr(),
                w1.data_ptr(), w2.data_ptr(), wd.data_ptr() if ds else 0,
                _ptr(p1) or 0, _ptr(p2) or 0, _ptr(pd) or 0, int(pm1), int(pm2), int(pmd), *bn,
                b0, b0 + row, b0 + 2 * row, (b0 + 4 * row) if ds else 0, (b0 + 5 * row) if ds else 0, b0 + 3 * row,
                s0, s0 + srow, (s0 + 2 * srow) if ds else 0, cws, cws)
            engine.check(L.lgs_block_forward(addr, _stream()))
        _written_by_engine(*touched)
        return buf[0], st[0], buf[1], buf[2], st[1], buf[3], (buf[4] if ds else None), (st[2] if ds else None)

    def block_backward(self, dy, saved, extra, kmap3, kmap1, pcs, params, relu_final, want_gin):
        """BasicBlock backward through lgs_block_backward -> the gradient tuple of models._BasicBlockFunction"""
        from . import modules as _modules
        from .modules import grad_slot_view
        L = engine.lib()
        self.block_calls = getattr(self, "block_calls", 0) + 1
        x, w1, g1, b1, w2, g2, b2, o1, st1, y1, o2, st2, y2 = saved
        wd, gd, bd, od, std = extra
        pw1, pg1, pb1, pw2, pg2, pb2, pwd, pgd, pbd = params
        n, cin = x.shape
        planes = w1.shape[2]
        dt = _dtype_code(x)
        dev = x.device
        ds = wd is not None
        esz = dy.element_size()
        if (dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) >= planes and (dy.stride(0) * esz) % 16 == 0
                and dy.data_ptr() % 16 == 0 and dy.dtype == x.dtype):
            dy_ld = dy.stride(0)
        else:
            dy, dy_ld = dy.contiguous(), planes
        pk = get_packed()
        with _dev(dev):
            p1, pm1 = pk.lookup(pcs[0], kmap3, 1, False, w1, w1, cin, planes, dt)
            p2, pm2 = pk.lookup(pcs[1], kmap3, 1, False, w2, w2, planes, planes, dt)
            pd, pmd = pk.lookup(pcs[2], kmap1, 1, False, wd, wd, cin, planes, dt) if ds else (None, 0)
            buf = torch.empty((5 if ds else 4, n, planes), dtype=x.dtype, device=dev)
            b0 = buf.data_ptr()
            row = n * planes * x.element_size()

            slots = [True]

            def wgrad_out(param, ref):
                v = grad_slot_view(param) if param is not None else None
                if v is None:
                    slots[0] = False
                    return torch.empty(ref.shape, dtype=torch.float32, device=dev)
                return v

            def affine_out(pg, pb, c):
                gv = grad_slot_view(pg) if pg is not None else None
                bv = grad_slot_view(pb) if pb is not None else None
                if gv is None or bv is None:
                    t = torch.empty((2, c), dtype=torch.float32, device=dev)
                    return t[0], t[1]
                return gv, bv
            gw1, gw2 = wgrad_out(pw1, w1), wgrad_out(pw2, w2)
            dg1, db1 = affine_out(pg1, pb1, planes)
            dg2, db2 = affine_out(pg2, pb2, planes)
            gwd = dgd = dbd = gind = None
            if ds:
                gwd = wgrad_out(pwd, wd)
                dgd, dbd = affine_out(pgd, pbd, planes)
                gind = torch.empty((n, cin), dtype=x.dtype, device=dev)
            cws = self._block_ws(L, kmap3, kmap1, cin, planes, dt, n, dev).data_ptr()
            # weight gradients beside the dgrad / BatchNorm chain (what modules.conv_weight_grad does call by call): every kernel
            # parameter owns a gradient-bucket slot and the batch is not one of the small ones that keep them on the compute stream
            side = None
            s_raw = s_ws = s_fork = e1 = e2 = ed = 0
            if not getattr(kmap3.mgr, "inline_wgrad", False) and slots[0] and _modules._DBG_WGRAD == "":
                side = self.side_stream(dev)
                s_raw = side.cuda_stream
                need = max(kmap3._ws_bytes(L, cin, planes, dt, 2), kmap3._ws_bytes(L, planes, planes, dt, 2),
                           kmap1._ws_bytes(L, cin, planes, dt, 2) if ds else 0)
                s_ws = _ws(need, dev, side).data_ptr()
                s_fork = _raw_event(self.fork_event(dev), side)
                e1, e2 = _raw_event(_param_event(pw1), side), _raw_event(_param_event(pw2), side)
                ed = _raw_event(_param_event(pwd), side) if ds else 0
            args_b, addr_b = self._blk_stage("b")
            engine.BLOCK_BWD_PACK.pack_into(
                args_b, 0, kmap3.h.value, kmap1.h.value if ds else 0, dt, int(relu_final), cin, planes, int(want_gin), 0,
                n, 0 if dy_ld == planes else dy_ld, dy.data_ptr(),
                x.data_ptr(), o1.data_ptr(), y1.data_ptr(), o2.data_ptr(), y2.data_ptr() if relu_final else 0, od.data_ptr() if ds else 0,
                st1.data_ptr(), st2.data_ptr(), std.data_ptr() if ds else 0,
                w1.data_ptr(), w2.data_ptr(), wd.data_ptr() if ds else 0,
                _ptr(p1) or 0, _ptr(p2) or 0, _ptr(pd) or 0, int(pm1), int(pm2), int(pmd),
                g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr(), gd.data_ptr() if ds else 0, bd.data_ptr() if ds else 0,
                b0, b0 + row, b0 + 2 * row, b0 + 3 * row, (b0 + 4 * row) if ds else 0, gind.data_ptr() if ds else 0,
                gw1.data_ptr(), gw2.data_ptr(), gwd.data_ptr() if ds else 0,
                dg1.data_ptr(), db1.data_ptr(), dg2.data_ptr(), db2.data_ptr(), dgd.data_ptr() if ds else 0, dbd.data_ptr() if ds else 0,
                cws, cws, s_raw, s_ws, s_fork, e1, e2, ed)
            engine.check(L.lgs_block_backward(addr_b, _stream()))
            if side is not None:
                # the side stream reads these after this call returns: their memory may only be reused in ITS order
                for t in (x, y1, buf):
                    t.record_stream(side)
                for p in (pw2, pw1, pwd) if ds else (pw2, pw1):       # the order the engine issued them in
                    _modules.note_side_wgrad(p)
        gin = (gind if ds else buf[1]) if want_gin else None
        # (all parameters of the fast path are fp32: the engine's fp32 gradients need no cast)
        out = (gin, None, None, None, gw1, dg1, db1, gw2, dg2, db2)
        if ds:
            out = out + (gwd, dgd, dbd)
        return out

    # ---- the same op in halves (SyncBN: statistics are exchanged between ranks in the middle)
    def bn_stats(self, x, conv_stats=None):
        """-> float32 [2C+1]: local mean, local M2, row count (the record one rank contributes to SyncBN's all-gather)"""
        L = engine.lib()
        x = x.contiguous()
        n, c = x.shape
        part, piv = conv_stats if conv_stats is not None else (None, None)
        with _dev(x.device):
            out = torch.empty(2 * c + 1, dtype=torch.float32, device=x.device)
            ws = _ws(_bn_ws_bytes(L, n, c), x.device)
            engine.check(L.lgs_bn_stats(_ptr(x), n, c, _ptr(out), _dtype_code(x), _ptr(ws), _ptr(part),
                                        int(part.shape[0]) if part is not None else 0, _ptr(piv), _stream()))
        return out

    def bn_sync_combine(self, all_stats, c, eps, momentum, running_mean, running_var, num_batches_tracked):
        """all_stats [world, 2C+1] -> (stats [2C] = global mean | invstd, inv_n [1] = 1 / global rows); one kernel"""
        L = engine.lib()
        world = all_stats.shape[0]
        with _dev(all_stats.device):
            stats = torch.empty(2 * c, dtype=torch.float32, device=all_stats.device)
            inv_n = torch.empty(1, dtype=torch.float32, device=all_stats.device)
            engine.check(L.lgs_bn_sync_combine(_ptr(all_stats), int(world), int(c), float(eps), float(momentum), _ptr(running_mean),
                                               _ptr(running_var), _ptr(num_batches_tracked), _ptr(stats), _ptr(inv_n), _stream()))
        _written_by_engine(running_mean, running_var, num_batches_tracked)
        return stats, inv_n

    def bn_apply(self, x, gamma, beta, stats, residual, relu, out_into=None):
        L = engine.lib()
        x = x.contiguous()
        n, c = x.shape
        with _dev(x.device):
            y, y_ld = self._y_out(x, out_into)
            if n == 0:                     # an empty batch (a zero-row tensor has no storage to point at): nothing to normalise
                return y
            res = residual.contiguous() if residual is not None else None
            engine.check(L.lgs_bn_apply(_ptr(x), n, c, _ptr(gamma), _ptr(beta), _ptr(stats), _ptr(res), int(relu), _ptr(y),
                                        _dtype_code(x), int(y_ld), _stream()))
        return y

    # ---- SyncBN as one call per direction on the engine's own RCCL communicator (csrc/lgs_comm.hip)
    @staticmethod
    def _y_out(x, out_into):
        """-> (y, row stride for the engine): a fresh tensor, or the column slice of a concat buffer (zero-copy ME.cat)"""
        if out_into is None:
            return torch.empty_like(x), 0
        buf, off = out_into
        n, c = x.shape
        y = buf[:, off:off + c]
        assert buf.dtype == x.dtype and buf.shape[0] == n and (buf.stride(0) * x.element_size()) % 16 == 0 and y.data_ptr() % 16 == 0
        return y, buf.stride(0)

    @staticmethod
    def _strided_in(t, c):
        """-> (tensor, row stride for the engine) of a [n, c] operand read in place when it is an aligned column slice, else a copy"""
        if t is None:
            return None, 0
        ld = _row_strided(t, c)
        if ld is not None:
            return t, ld
        return t.contiguous(), 0

    def bn_forward_sync(self, comm, x, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, residual, relu,
                        out_into=None):
        """-> y, stats [2C], inv_n [1]"""
        L = engine.lib()
        x = x.contiguous()
        n, c = x.shape
        with _dev(x.device):
            y, y_ld = self._y_out(x, out_into)
            stats = torch.empty(2 * c + 1, dtype=torch.float32, device=x.device)      # [mean | invstd | 1 / global rows]
            res = residual.contiguous() if residual is not None else None
            ws = _ws(L.lgs_bn_sync_workspace_bytes(n, c, comm.world), x.device)
            engine.check(L.lgs_bn_forward_sync(comm.h, _ptr(x), n, c, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                                               _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _ptr(res), int(relu),
                                               _ptr(y), stats.data_ptr(), stats.data_ptr() + 8 * c, _dtype_code(x), _ptr(ws), int(y_ld), _stream()))
        _written_by_engine(running_mean, running_var, num_batches_tracked)
        return y, stats[:2 * c], stats[2 * c:]

    def bn_backward_sync(self, comm, x, y, dy, gamma, beta, stats, inv_n, relu, want_residual, dgamma_out=None, dbeta_out=None):
        """-> dx, dres (dgamma_out / dbeta_out receive the LOCAL parameter gradients)"""
        L = engine.lib()
        n, c = x.shape
        dy, dy_ld = self._strided_in(dy, c)
        y, y_ld = self._strided_in(y, c)
        with _dev(x.device):
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_residual else None
            ws = _ws(L.lgs_bn_sync_workspace_bytes(n, c, comm.world), x.device)
            engine.check(L.lgs_bn_backward_sync(comm.h, _ptr(x), _ptr(y), _ptr(dy), n, c, _ptr(gamma), _ptr(beta), _ptr(stats), _ptr(inv_n),
                                                int(relu), _ptr(dx), _ptr(dres), _ptr(dgamma_out), _ptr(dbeta_out), _dtype_code(x),
                                                _ptr(ws), int(dy_ld), int(y_ld), _stream()))
        return dx, dres

    def bn_backward_reduce(self, x, y, dy, gamma, beta, stats, relu, dgamma_out=None, dbeta_out=None):
        """-> sums [2C] (local sum dy', sum dy' xhat); the same vectors are also written to dgamma_out / dbeta_out"""
        L = engine.lib()
        n, c = x.shape
        dy, dy_ld = self._strided_in(dy, c)
        y, y_ld = self._strided_in(y, c)
        with _dev(x.device):
            if n == 0:                     # empty batch: zero sums (and zero parameter gradients)
                for t in (dgamma_out, dbeta_out):
                    if t is not None:
                        t.zero_()
                return torch.zeros(2 * c, dtype=torch.float32, device=x.device)
            sums = torch.empty(2 * c, dtype=torch.float32, device=x.device)
            ws = _ws(_bn_ws_bytes(L, n, c), x.device)
            engine.check(L.lgs_bn_backward_reduce(_ptr(x), _ptr(y), _ptr(dy), n, c, _ptr(gamma), _ptr(beta), _ptr(stats), int(relu), _ptr(sums),
                                                  _ptr(dgamma_out), _ptr(dbeta_out), _dtype_code(x), _ptr(ws), int(dy_ld), int(y_ld), _stream()))
        return sums

    def bn_backward_apply(self, x, y, dy, gamma, beta, stats, sums, inv_n_total, relu, want_residual):
        """inv_n_total: python float, or a device scalar tensor (1 / global row count, no host sync)"""
        L = engine.lib()
        n, c = x.shape
        dev_inv = inv_n_total if torch.is_tensor(inv_n_total) else None
        dy, dy_ld = self._strided_in(dy, c)
        y, y_ld = self._strided_in(y, c)
        with _dev(x.device):
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_residual else None
            if n == 0:
                return dx, dres
            engine.check(L.lgs_bn_backward_apply(_ptr(x), _ptr(y), _ptr(dy), n, c, _ptr(gamma), _ptr(beta), _ptr(stats), _ptr(sums),
                                                 0.0 if dev_inv is not None else float(inv_n_total), _ptr(dev_inv), int(relu), _ptr(dx),
                                                 _ptr(dres), _dtype_code(x), int(dy_ld), int(y_ld), _stream()))
        return dx, dres

    # ---- CLIP contraction: lgs_clip_similarity
    def clip_similarity(self, feats, anchors):
        _require_dev(feats, "features")
        L = engine.lib()
        feats = feats.contiguous()
        anchors = anchors.detach().contiguous().float()
        n, c = feats.shape
        na = anchors.shape[0]
        dt = _dtype_code(feats)
        with _dev(feats.device):
            sim = torch.empty((n, na), dtype=torch.float32, device=feats.device)
            inv = torch.empty(max(n, 1), dtype=torch.float32, device=feats.device)
            ws = _ws(L.lgs_clip_workspace_bytes(c, na, dt), feats.device)
            engine.check(L.lgs_clip_similarity(_ptr(feats), n, c, _ptr(anchors), na, _ptr(sim), _ptr(inv), dt, _ptr(ws),
                                               _stream()))
        return sim, inv[:n]

    # ---- fused CLIP text-anchor loss: lgs_clip_loss_forward / lgs_clip_loss_backward
    CLIP_LOSS_MAX_ANCHORS = 224

    def clip_loss_forward(self, feats, anchors, labels, neg, ignore_label, want_sim=False):
        """one pass over the features -> (d_pos [N], d_neg [N], pred [N] int64, saved-for-backward tuple, sim or None)"""
        _require_dev(feats, "features")
        L = engine.lib()
        feats = feats.contiguous()
        anchors = anchors.detach().contiguous().float()
        labels = labels.contiguous().to(torch.int64)
        neg = neg.contiguous().to(torch.int64)
        n, c = feats.shape
        na, k = anchors.shape[0], neg.shape[1]
        dt = _dtype_code(feats)
        dev = feats.device
        with _dev(dev):
            d_pos = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            d_neg = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            inv = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            pred = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
            tn = torch.empty((na, c), dtype=torch.float32, device=dev)
            sim = torch.empty((n, na), dtype=torch.float32, device=dev) if want_sim else None
            ws = _ws(L.lgs_clip_loss_workspace_bytes(c, na, dt), dev)
            engine.check(L.lgs_clip_loss_forward(_ptr(feats), n, c, _ptr(anchors), na, _ptr(labels), _ptr(neg), int(k),
                                                 int(ignore_label), _ptr(d_pos), _ptr(d_neg), _ptr(pred), _ptr(inv), _ptr(tn),
                                                 _ptr(sim), dt, _ptr(ws), _stream()))
        return d_pos[:n], d_neg[:n], pred[:n], (feats, tn, labels, neg, inv), sim

    def clip_loss_backward(self, saved, d_pos, d_neg, g_dpos, g_dneg, ignore_label):
        L = engine.lib()
        feats, tn, labels, neg, inv = saved
        n, c = feats.shape
        with _dev(feats.device):
            gf = torch.empty_like(feats)
            gp = g_dpos.contiguous().float() if g_dpos is not None else None
            gn = g_dneg.contiguous().float() if g_dneg is not None else None
            engine.check(L.lgs_clip_loss_backward(_ptr(feats), n, c, _ptr(tn), tn.shape[0], _ptr(labels), _ptr(neg), int(neg.shape[1]),
                                                  int(ignore_label), _ptr(inv), _ptr(d_pos), _ptr(d_neg), _ptr(gp), _ptr(gn), _ptr(gf),
                                                  _dtype_code(feats), _stream()))
        return gf

    def clip_loss_backward_anchors(self, saved, g_dpos, g_dneg, ignore_label):
        """-> d loss / d t^ [n_anchor, c] float32 (gradient w.r.t. the NORMALISED anchors): lgs_clip_loss_backward_anchors"""
        L = engine.lib()
        feats, tn, labels, neg, inv = saved
        n, c = feats.shape
        na = tn.shape[0]
        a8 = (na + 7) // 8 * 8
        dt = _dtype_code(feats)
        with _dev(feats.device):
            gt = torch.empty((c, a8), dtype=torch.float32, device=feats.device)
            gp = g_dpos.contiguous().float() if g_dpos is not None else None
            gn = g_dneg.contiguous().float() if g_dneg is not None else None
            ws = _ws(L.lgs_clip_anchor_grad_workspace_bytes(n, c, na, dt), feats.device)
            engine.check(L.lgs_clip_loss_backward_anchors(_ptr(feats), n, c, na, _ptr(labels), _ptr(neg), int(neg.shape[1]),
                                                          int(ignore_label), _ptr(inv), _ptr(gp), _ptr(gn), _ptr(gt), dt, _ptr(ws),
                                                          _stream()))
        return gt[:, :na].t()

    # ---- fused softmax cross-entropy: lgs_ce_forward_backward
    def cross_entropy(self, logits, labels, ignore_index, want_grad=True, grad_scale=None, inv_valid=None):
        """mean CE over the non-ignored rows.  want_grad=False: loss only; grad_scale (device scalar): gradient only,
        already multiplied by it (the two halves of one kernel, so the upstream gradient never needs its own pass).
        inv_valid: 1 / #counted rows from an earlier call on the same labels (the forward's, reused by the backward).
        -> (loss, dlogits, inv_valid)"""
        _require_dev(logits, "logits")
        L = engine.lib()
        logits = logits.contiguous()
        labels = labels.contiguous().to(torch.int64)
        n, c = logits.shape
        dt = _dtype_code(logits)
        if n == 0:      # empty batch: loss 0 (the convention of the all-ignored batch), an empty gradient
            one = self._one(logits.device)
            return (one * 0.0 if grad_scale is None else None), (torch.empty_like(logits) if (want_grad or grad_scale is not None) else None), \
                (inv_valid if inv_valid is not None else one)
        with _dev(logits.device):
            if inv_valid is None:
                # the same predicate the kernel uses: a label outside [0, C) is an ignored row, not a counted one
                cnt = torch.empty(1, dtype=torch.int32, device=logits.device)
                engine.check(L.lgs_ce_count_valid(_ptr(labels), n, c, int(ignore_index), _ptr(cnt), _stream()))
                inv_valid = cnt.to(torch.float32).clamp_min_(1.0).reciprocal_().reshape(())
            scale = inv_valid
            if grad_scale is not None:
                scale = scale * grad_scale.to(torch.float32).reshape(())
            loss_rows = torch.empty(max(n, 1), dtype=torch.float32, device=logits.device) if grad_scale is None else None
            dlogits = torch.empty_like(logits) if (want_grad or grad_scale is not None) else None
            engine.check(L.lgs_ce_forward_backward(_ptr(logits), n, c, _ptr(labels), int(ignore_index), _ptr(scale),
                                                   _ptr(loss_rows), _ptr(dlogits), dt, _stream()))
        loss = loss_rows[:n].sum() * inv_valid if loss_rows is not None else None
        return loss, dlogits, inv_valid

    def cross_entropy_rows(self, logits, labels, ignore_index, row_grad=None):
        """nn.CrossEntropyLoss(reduction='none') (pl_BaselineTrainer.py:94 under balanced_category_sampling): row_grad=None ->
        the per-row losses [N] (0 for ignored rows); row_grad [N] fp32 = the upstream gradient -> d(logits), one pass either way
        (lgs_ce_forward_backward_rows).  No denominator: the caller's reduction owns it."""
        _require_dev(logits, "logits")
        L = engine.lib()
        logits = logits.contiguous()
        labels = labels.contiguous().to(torch.int64)
        n, c = logits.shape
        if n == 0:
            return torch.empty(0, dtype=torch.float32, device=logits.device) if row_grad is None else torch.empty_like(logits)
        with _dev(logits.device):
            one = self._one(logits.device)
            if row_grad is None:
                loss_rows = torch.empty(n, dtype=torch.float32, device=logits.device)
                engine.check(L.lgs_ce_forward_backward_rows(_ptr(logits), n, c, _ptr(labels), int(ignore_index), _ptr(one), None,
                                                            _ptr(loss_rows), None, _dtype_code(logits), _stream()))
                return loss_rows
            row_grad = row_grad.contiguous().to(torch.float32)
            dlogits = torch.empty_like(logits)
            engine.check(L.lgs_ce_forward_backward_rows(_ptr(logits), n, c, _ptr(labels), int(ignore_index), _ptr(one), _ptr(row_grad),
                                                        None, _ptr(dlogits), _dtype_code(logits), _stream()))
            return dlogits

    def split_stats(self, loss_rows, labels, group_of_class, ignore_index):
        """-> [3, 2] fp32: (sum of loss_rows, number of rows) of the head / common / tail points (lgs_split_stats), no host sync"""
        _require_dev(loss_rows, "loss_rows")
        L = engine.lib()
        loss_rows = loss_rows.detach().contiguous().to(torch.float32)
        labels = labels.contiguous().to(torch.int64)
        group_of_class = group_of_class.contiguous().to(torch.int32)
        n = loss_rows.shape[0]
        rows = max(1, min(1024, (n + 2047) // 2048))
        with _dev(loss_rows.device):
            partial = torch.empty(rows, 6, dtype=torch.float32, device=loss_rows.device)
            engine.check(L.lgs_split_stats(_ptr(loss_rows), _ptr(labels), n, _ptr(group_of_class), int(group_of_class.shape[0]),
                                           int(ignore_index), _ptr(partial), rows, _stream()))
        return partial.sum(0).view(3, 2)

    def _one(self, device):
        """a device-resident 1.0f (the kernels take their scalar factors from device memory)"""
        cache = self.__dict__.setdefault("_one_cache", {})
        t = cache.get(device.index)
        if t is None:
            t = cache[device.index] = torch.ones((), dtype=torch.float32, device=device)
        return t
