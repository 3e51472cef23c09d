"""Data-parallel helpers: one process per GPU, RCCL over xGMI through torch.distributed.

Replaces (functionally) what the reference gets from PyTorch-Lightning's DDPPlugin + ME SyncBatchNorm:
  /root/reference/main.py:121-123   MinkowskiSyncBatchNorm.convert_sync_batchnorm when num_gpu > 1
  /root/reference/main.py:192-195   DDPPlugin(find_unused_parameters=True) -> NCCL gradient all-reduce
Scenes are independent: every rank owns its own batch, coordinate manager and kernel maps;
the only exchange per step is the gradient all-reduce (+ SyncBN statistics).

Design for xGMI (point-to-point links, ring collectives are per-link bound): gradients live in a few
large flat fp32 buckets (param.grad are views into them, so there is no copy-in/copy-out), each bucket is
all-reduced asynchronously on RCCL's stream the moment its last gradient has been accumulated, i.e.
overlapped with the rest of backward; finalize() waits once before the optimiser step.
"""
import torch
import torch.distributed as dist

_TIMING = {"on": False}
_SYNCBN_EVENTS = []      # (start, end) HIP events around SyncBN collectives while BucketedDDP.enable_timing() is active


def _timed_collective(fn, tensor_is_cuda):
    """run one blocking SyncBN collective; with timing on, bracket it with events on the compute stream"""
    if _TIMING["on"] and tensor_is_cuda:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        _SYNCBN_EVENTS.append((a, b))
    else:
        fn()


class BucketedDDP:
    """force_collectives=True issues the all-reduces even with a world of one rank (used by the single-GPU RCCL test:
    the collective path, its stream ordering and the bucket views are then the ones an 8-GPU run executes)."""

    def __init__(self, module, bucket_mb=32.0, process_group=None, force_collectives=False, allreduce="ring"):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.reduce = self.world > 1 or (force_collectives and dist.is_initialized())
        self._defer = False
        params = [p for p in module.parameters() if p.requires_grad]
        # gradients become ready roughly in reverse registration order
        params = list(reversed(params))
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []  # dict(flat, params, pending, launched, views)
        cur, cur_n = [], 0
        for p in params:
            if cur and cur_n + p.numel() > cap:
                self._make_bucket(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._make_bucket(cur)
        self._handles = []
        self._next = 0                  # first bucket whose collective has not been issued this step
        self._step = 0                  # bumped by zero_grad(): wgrad events of earlier steps are never waited on
        self.allreduce = allreduce      # "ring": dist.all_reduce per bucket; "rs_ag": reduce-scatter + all-gather on the flat bucket
        self.timing = None              # set by enable_timing(): HIP events around collectives / waits (bench.py --gpus N)
        if self.reduce:
            for p in params:
                dist.broadcast(p.data, src=0, group=self.group)
            for b in module.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b.data, src=0, group=self.group)
            # the broadcast wrote through `.data`: packed weight images made by a warm-up forward are stale now
            from .me.core import get_backend
            be = get_backend()
            if hasattr(be, "invalidate_packed_weights"):
                be.invalidate_packed_weights()

    def _make_bucket(self, params):
        n = sum(p.numel() for p in params)
        # padded so that the reduce-scatter / all-gather halves ("rs_ag") split evenly and stay 16-byte aligned per rank
        # layout: [gradients | one "used" flag per parameter | padding].  The flags ride along in the same collective: after
        # the reduction flag > 0 <=> SOME rank produced a gradient for that parameter this step, which is what the optimiser
        # must key on (a parameter unused on this rank but used on another still has to move identically everywhere)
        q = 4 * max(self.world, 1)
        flat = torch.zeros((n + len(params) + q - 1) // q * q, dtype=torch.float32, device=params[0].device)
        off = 0
        bucket = {"flat": flat, "params": params, "pending": len(params), "n": len(params), "views": [], "launched": False,
                  "flag_off": n, "grad_elems": n}
        flat[n:n + len(params)] = 1.0
        for p in params:
            # the engine's backward kernels write gradients straight into this slot (me.modules.grad_slot_view);
            # any other producer falls back to autograd's own accumulation into the same memory
            p._lgs_grad_slot = (flat, off, tuple(p.shape))
            p._lgs_ddp = self
            bucket["views"].append((p, off))
            off += p.numel()
            if self.reduce:
                p.register_post_accumulate_grad_hook(self._hook(bucket))
        self.buckets.append(bucket)

    def _side_streams(self):
        from .me.core import get_backend
        be = get_backend()
        return list(getattr(be, "_side", {}).values())

    def _join_side(self):
        """weight gradients are produced on the backend's side stream: order them before any consumer"""
        if self.buckets and self.buckets[0]["flat"].is_cuda:
            cur = torch.cuda.current_stream()
            for s in self._side_streams():
                cur.wait_stream(s)

    def _hook(self, bucket):
        def fn(param):
            if self._defer:                      # no_sync(): gradients accumulate locally, nothing is reduced
                return
            bucket["pending"] -= 1
            if bucket["pending"] == 0:
                self._launch_ready()
            elif bucket["pending"] < 0:
                raise RuntimeError("BucketedDDP: a gradient arrived after its bucket was reduced -- call zero_grad() before "
                                   "every backward, or wrap all but the last micro-batch of an accumulation in no_sync()")
        return fn

    def _launch_ready(self):
        """Collectives are issued in FIXED bucket order on every rank: bucket i goes out only once buckets 0..i-1 have.
        Ranks whose sets of unused parameters differ (a data-dependent loss branch -- the reference runs
        find_unused_parameters=True for that reason) would otherwise launch bucket k from a hook on one rank and from
        finalize() on another: mismatched all-reduce sequences hang RCCL or silently mix buckets."""
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _collect(self, bucket):
        """Every parameter's gradient must live in its slot of the flat bucket before the bucket is reduced / handed to
        the optimiser.  Gradients the engine wrote there directly (conv weights on the side stream, BN affine) already
        do; anything autograd produced elsewhere (torch modules, the engine's non-slot fallbacks, a cloned view) is
        copied in and `.grad` re-pointed at the slot.  The check is one pointer compare per parameter on the host,
        every step: a cached list of "stray" parameters went stale whenever a producer changed its mind after step 1."""
        base = bucket["flat"].data_ptr()
        for p, off in bucket["views"]:
            g = p.grad
            if g is not None and g.data_ptr() != base + off * 4:
                bucket["flat"][off:off + p.numel()].copy_(g.reshape(-1))
                p.grad = bucket["flat"][off:off + p.numel()].view_as(p)

    def _wait_bucket_wgrads(self, bucket):
        """order the current stream after the side-stream weight gradients of THIS bucket only (one event per conv
        parameter, recorded behind its wgrad kernels by MinkowskiConvolutionFunction.backward): a bucket that is ready
        early must not wait for weight gradients of later layers that merely sit in front of it in the side stream's
        queue -- nor for the whole stream, as a stream-wide join does"""
        if not bucket["flat"].is_cuda:
            return
        cur = torch.cuda.current_stream()
        last = None
        for p, _ in bucket["views"]:
            ev = getattr(p, "_lgs_wgrad_event", None)
            if ev is not None and getattr(p, "_lgs_wgrad_step", -1) == self._step:
                # the side stream runs in order: the event recorded LAST covers the earlier ones
                if last is None or p._lgs_wgrad_seq > last[0]:
                    last = (p._lgs_wgrad_seq, ev)
        if last is not None:
            cur.wait_event(last[1])

    def _launch(self, bucket):
        self._wait_bucket_wgrads(bucket)
        self._collect(bucket)
        bucket["launched"] = True
        flat = bucket["flat"]
        # "used" flags: all ones unless this rank produced no gradient for a parameter (then the bucket was not launched
        # from a hook but flushed by finalize(), so this host-side test is rare and never on the overlapped path)
        fo = bucket["flag_off"]
        for i, (p, _) in enumerate(bucket["views"]):
            if p.grad is None:
                flat[fo + i] = 0.0
        if self.world > 1:
            flat.div_(self.world)
        t = self.timing
        if t is not None and flat.is_cuda:
            e = torch.cuda.Event(enable_timing=True); e.record(); t["launch_ev"].append(e)
        if self.allreduce == "rs_ag":
            # reduce-scatter + all-gather on the flat bucket (padded to a multiple of the world size): every rank reduces
            # 1/W of the bucket and the two halves move (W-1)/W of the bytes each over ALL peers' links at once, instead of
            # one ring's per-link rate (38 MB per link vs 265 MB over one)
            W = self.world
            chunk = flat.numel() // W
            r = dist.get_rank(self.group)
            mine = flat[r * chunk:(r + 1) * chunk]
            if dist.get_backend(self.group) == "nccl":
                dist.reduce_scatter_tensor(mine, flat, group=self.group, async_op=True)          # in place: out = in + r * chunk
                self._handles.append(dist.all_gather_into_tensor(flat, mine, group=self.group, async_op=True))
            else:
                # gloo (CPU / single-GPU dry runs of the N>1 logic) has neither collective: W rooted reduces + an all-gather
                # exercise the same partitioning
                for j in range(W):
                    dist.reduce(flat[j * chunk:(j + 1) * chunk], dst=dist.get_global_rank(self.group, j) if self.group is not None else j,
                                group=self.group)
                parts = [torch.empty_like(mine) for _ in range(W)]
                dist.all_gather(parts, mine.clone(), group=self.group)
                for j in range(W):
                    flat[j * chunk:(j + 1) * chunk].copy_(parts[j])
        else:
            self._handles.append(dist.all_reduce(flat, group=self.group, async_op=True))

    def enable_timing(self):
        """bench.py --gpus N: HIP events on the compute stream around what the data-parallel path adds to a step, so that
        a first multi-GPU run can be read: `allreduce_wait` = time finalize() stalls the compute stream for bucket
        collectives that backward did not hide; `syncbn` = compute-stream stalls inside the SyncBN collectives."""
        self.timing = {"launch_ev": [], "wait": [], "syncbn": _SYNCBN_EVENTS}
        _SYNCBN_EVENTS.clear()
        _TIMING["on"] = True

    def timing_summary(self, steps):
        """-> dict of per-step milliseconds (call after torch.cuda.synchronize())"""
        t = self.timing
        if t is None:
            return None
        out = {"allreduce_exposed_wait_ms": sum(a.elapsed_time(b) for a, b in t["wait"]) / max(steps, 1),
               "bucket_collectives_per_step": len(t["launch_ev"]) / max(steps, 1),
               # (collectives issued by the engine inside its own call are counted, not timed: entries without events)
               "syncbn_collective_ms": sum(ab[0].elapsed_time(ab[1]) for ab in t["syncbn"] if ab is not None) / max(steps, 1),
               "syncbn_collectives_in_engine_calls_per_step": sum(1 for ab in t["syncbn"] if ab is None) / max(steps, 1),
               "syncbn_collectives_per_step": len(t["syncbn"]) / max(steps, 1)}
        t["launch_ev"].clear(); t["wait"].clear(); t["syncbn"].clear()
        return out

    def zero_grad(self):
        self._next = 0
        self._step += 1
        # two multi-tensor launches for all buckets (a zero_ and a slice fill per bucket were 11 launches at the head of every step)
        flats = [b["flat"] for b in self.buckets]
        if flats:
            torch._foreach_zero_(flats)
            flags = getattr(self, "_flag_views", None)
            if flags is None or len(flags) != len(flats) or any(f._base is not b["flat"] for f, b in zip(flags, self.buckets)):
                flags = self._flag_views = [b["flat"][b["flag_off"]:b["flag_off"] + b["n"]] for b in self.buckets]
            torch._foreach_add_(flags, 1.0)          # (just zeroed: the "used" flags are all ones again)
        for b in self.buckets:
            b["pending"] = b["n"]
            b["launched"] = False
            for p, _ in b["views"]:
                p.grad = None

    class _NoSync:
        def __init__(self, ddp):
            self.ddp = ddp

        def __enter__(self):
            self.prev, self.ddp._defer = self.ddp._defer, True

        def __exit__(self, *exc):
            self.ddp._defer = self.prev
            # the micro-batches inside the context wrote weight gradients on the side stream; the next backward ACCUMULATES
            # into the same slots on the main stream (p.grad is set, so grad_slot_view declines): order it behind them
            self.ddp._join_side()

    def no_sync(self):
        """Gradient accumulation (the reference's insseg trainer supports iter_size > 1): backward passes inside the
        context only accumulate into the flat buckets; the first backward outside it reduces them."""
        return BucketedDDP._NoSync(self)

    def finalize(self):
        """call after backward(): waits for the side-stream weight gradients and the in-flight bucket all-reduces
        (and flushes buckets whose parameters received no gradient this step -- the reference runs
        find_unused_parameters=True).  Collectives go out in fixed bucket order on every rank (_launch_ready): whatever
        the hooks could not issue yet is issued here, in the same order.  Inside no_sync() nothing is reduced."""
        self._join_side()
        if self.reduce and not self._defer:
            while self._next < len(self.buckets):
                self._launch(self.buckets[self._next])
                self._next += 1
            t = self.timing
            if t is not None and self.buckets and self.buckets[0]["flat"].is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for h in self._handles:
                    h.wait()
                b.record()
                t["wait"].append((a, b))
            else:
                for h in self._handles:
                    h.wait()
            for b in self.buckets:      # ready for the next backward even if the caller clears grads some other way
                b["pending"] = b["n"]
                b["launched"] = False
            self._next = 0
        else:
            for b in self.buckets:
                self._collect(b)
        self._handles = []

    def __call__(self, *a, **k):
        return self.module(*a, **k)


class FlatSGD:
    """SGD with momentum / dampening / weight decay (the reference's optimiser, /root/reference/lib/solvers.py:
    SGD(momentum=0.9, dampening=0.1, weight_decay=1e-4)) applied to the FLAT gradient buckets of a BucketedDDP:
    parameters are re-homed as views of flat buffers too, so one step is four elementwise kernels per bucket
    instead of a multi-tensor pass over ~190 tensors (whose host-side preparation left the GPU idle ~1 ms per step).
    Same update rule as torch.optim.SGD; parameters that received no gradient are left untouched."""

    def __init__(self, ddp, lr, momentum=0.0, dampening=0.0, weight_decay=0.0):
        self.ddp, self.lr, self.momentum, self.dampening, self.weight_decay = ddp, lr, momentum, dampening, weight_decay
        self.state = []
        for b in ddp.buckets:
            flat_p = torch.zeros_like(b["flat"])      # (the bucket's padding tail stays zero)
            for p, off in b["views"]:
                flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + p.numel()].view_as(p)
            self.state.append({"p": flat_p, "buf": None, "mask": torch.ones_like(flat_p)})
        self.steps = 0

    @torch.no_grad()
    def step(self):
        # a pending ME result held across the update (me/deferred.py) was recorded with the OLD weights: it runs before they move
        from .me import deferred as _deferred
        _deferred.flush_all()
        fused = self.state and self.state[0]["p"].is_cuda
        for b, st in zip(self.ddp.buckets, self.state):
            g = b["flat"]
            # parameters without a gradient this step (grad is None) must not move, not even by weight decay / momentum
            unused = [(i, p, off) for i, (p, off) in enumerate(b["views"]) if p.grad is None]
            if unused:
                st["mask"].fill_(1.0)
                reduced = self.ddp.reduce and self.ddp.world > 1
                for i, p, off in unused:
                    if reduced:
                        # unused HERE; another rank may have used it: the all-reduced "used" flag decides, on the device (no
                        # host sync), so every rank applies the same update
                        st["mask"][off:off + p.numel()] = (g[b["flag_off"] + i] > 0).to(torch.float32)
                    else:
                        st["mask"][off:off + p.numel()] = 0.0
            ne = b["grad_elems"]          # the "used" flags and the padding behind the gradients are not parameters
            if fused:   # one kernel per bucket (lgs_sgd_step) instead of four elementwise passes
                import ctypes
                from . import engine
                first = st["buf"] is None
                if first and self.momentum != 0:
                    st["buf"] = torch.zeros_like(g)
                vp = ctypes.c_void_p
                with torch.cuda.device(g.device):
                    engine.check(engine.lib().lgs_sgd_step(
                        vp(st["p"].data_ptr()), vp(g.data_ptr()), vp(st["buf"].data_ptr()) if st["buf"] is not None else vp(None),
                        vp(st["mask"].data_ptr()) if unused else vp(None), int(ne), float(self.lr), float(self.momentum),
                        float(self.dampening), float(self.weight_decay), int(first), vp(torch.cuda.current_stream(g.device).cuda_stream)))
                continue
            d = g.add(st["p"], alpha=self.weight_decay) if self.weight_decay != 0 else g.clone()
            d[ne:] = 0.0
            if self.momentum != 0:
                if st["buf"] is None:
                    st["buf"] = d.clone()
                else:
                    st["buf"].mul_(self.momentum).add_(d, alpha=1 - self.dampening)
                d = st["buf"]
            if unused:
                st["p"].addcmul_(d, st["mask"], value=-self.lr)
            else:
                st["p"].add_(d, alpha=-self.lr)
        self.steps += 1
        if fused:
            # the parameters were updated through the C-ABI (no torch version bump): packed weight images are stale
            from .me.core import get_backend
            be = get_backend()
            if hasattr(be, "weights_updated"):
                be.weights_updated()


class _SyncBNFunction(torch.autograd.Function):
    """Batch statistics over the rows of ALL ranks with one packed all-reduce per direction:
    forward [sum | sumsq | count] (2C+1 floats), backward [sum dy | sum dy*xhat] (2C floats)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        xf = x.float()
        c = xf.shape[1]
        packed = torch.empty(2 * c + 1, dtype=torch.float32, device=x.device)
        packed[:c] = xf.sum(0)
        packed[c:2 * c] = (xf * xf).sum(0)
        packed[2 * c] = float(xf.shape[0])
        dist.all_reduce(packed, group=group)
        n = packed[2 * c]
        mean = packed[:c] / n
        var = (packed[c:2 * c] / n - mean * mean).clamp_min(0)
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean * momentum)
                running_var.mul_(1 - momentum).add_(var * (n / (n - 1).clamp_min(1)) * momentum)
        xhat = (xf - mean) * invstd
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.group = group
        return (xhat * weight.float() + bias.float()).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, invstd, n = ctx.saved_tensors
        g = dy.float()
        c = g.shape[1]
        packed = torch.empty(2 * c, dtype=torch.float32, device=g.device)
        packed[:c] = g.sum(0)
        packed[c:] = (g * xhat).sum(0)
        dbeta, dgamma = packed[:c].clone(), packed[c:].clone()   # parameter grads stay local (DDP averages them)
        dist.all_reduce(packed, group=ctx.group)
        dx = (g - packed[:c] / n - xhat * (packed[c:] / n)) * (weight.float() * invstd)
        return dx.to(dy.dtype), dgamma.to(weight.dtype), dbeta.to(weight.dtype), None, None, None, None, None


class EngineComm:
    """The engine's own RCCL communicator for one process group (csrc/lgs_comm.hip): SyncBN's per-layer collectives are issued by
    the engine ON THE COMPUTE STREAM between its kernels instead of through ProcessGroupNCCL (its stream hand-overs and ~60 us of
    host work per collective).  Created COLLECTIVELY the first time a SyncBN layer of the group runs: rank 0 draws the id, one
    torch.distributed broadcast shares it, every rank initialises, and one all-reduce(MIN) of a success flag over the torch group
    decides for ALL ranks together -- a rank that failed anywhere on the way (no librccl, id, init) still takes part in both
    collectives, so no rank is left waiting and no rank uses the communicator unless every rank has it (advisor, round 4).

    Default (`LGS_SYNCBN_ENGINE_COMM=-1`, "auto"): used in a world of ONE rank only (the single-GPU measurement of the per-rank
    path, bench.py dp_path_world1); with more than one rank SyncBN keeps torch.distributed's collectives -- the engine
    communicator's kernels on the compute stream would be in flight next to ProcessGroupNCCL's bucket all-reduces on its stream,
    two communicators at once, which this build has never been able to execute with a peer (one GPU per box).  `=1` turns it on
    for any world (what a first multi-GPU bring-up should A/B), `=0` off."""
    _by_group = {}

    def __init__(self, group, device, ipc=False):
        import ctypes
        from . import engine
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.h, self.world, self.rank, self._L = None, world, rank, None
        self.ipc = ipc
        if ipc:
            self._init_ipc(group, device, world, rank)
            return
        err = None
        buf = ctypes.create_string_buffer(128)
        try:
            L = self._L = engine.lib()
            if rank == 0:
                engine.check(L.lgs_comm_unique_id(buf))
        except Exception as e:
            err = e
        box = [bytes(buf.raw) if err is None else b""]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if err is None and len(box[0]) == 128:
            try:
                h = ctypes.c_void_p(None)
                idx = device.index if device.index is not None else torch.cuda.current_device()
                engine.check(L.lgs_comm_create(ctypes.create_string_buffer(box[0], 128), world, rank, idx, ctypes.byref(h)))
                self.h = h
            except Exception as e:
                err = e
        elif err is None:
            err = RuntimeError("rank 0 could not draw an RCCL unique id")
        ok = torch.tensor([1 if err is None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        self.ok = bool(int(ok.item()))
        self.error = err
        if not self.ok:
            self.close()

    def _init_ipc(self, group, device, world, rank):
        """mailbox mode (knob SYNCBN_IPC, csrc/lgs_comm.hip): every rank allocates its mailbox, the 64-byte IPC handles go round in
        ONE all-gather over the torch group (any backend), every rank maps the others'.  Failure anywhere is agreed on collectively,
        as for the RCCL communicator."""
        import ctypes
        from . import engine
        err, mine = None, bytes(64)
        try:
            L = self._L = engine.lib()
            h, buf = ctypes.c_void_p(None), ctypes.create_string_buffer(64)
            idx = device.index if device.index is not None else torch.cuda.current_device()
            engine.check(L.lgs_comm_create_ipc(world, rank, idx, ctypes.byref(h), buf))
            self.h, mine = h, bytes(buf.raw)
        except Exception as e:
            err = e
        box = [None] * world
        dist.all_gather_object(box, mine if err is None else b"", group=group)
        if err is None and all(isinstance(b, bytes) and len(b) == 64 for b in box):
            try:
                engine.check(self._L.lgs_comm_ipc_open(self.h, ctypes.create_string_buffer(b"".join(box), 64 * world)))
            except Exception as e:
                err = e
        elif err is None:
            err = RuntimeError("a peer could not export its mailbox")
        ok = torch.tensor([1 if err is None else 0], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        self.ok = bool(int(ok.item()))
        self.error = err
        if not self.ok:
            self.close()

    def close(self):
        if self.h is not None and self.h.value:
            self._L.lgs_comm_destroy(self.h)
        self.h = None

    @classmethod
    def get(cls, group, device):
        """-> EngineComm of this group, or None (off by knob, not an RCCL group, or some rank could not create it: said once)"""
        from . import tuning as _tuning
        g = group if group is not None else dist.group.WORLD
        key = (g, device.index)                    # the group OBJECT (kept alive by the key): id() of a freed group can be reused
        if key in cls._by_group:
            return cls._by_group[key]
        comm = None
        knob = _tuning.host("SYNCBN_ENGINE_COMM")
        want = knob == 1 or (knob < 0 and dist.get_world_size(group) == 1)
        ipc = _tuning.host("SYNCBN_IPC") != 0
        if ipc or (want and dist.get_backend(group) == "nccl"):
            c = cls(group, device, ipc=ipc)
            if c.ok:
                comm = c
            else:
                import sys
                print("[lgs] engine-side %s unavailable on at least one rank (this rank: %s): SyncBN keeps "
                      "torch.distributed's collectives on every rank" % ("mailbox exchange" if ipc else "RCCL communicator", c.error), file=sys.stderr)
        cls._by_group[key] = comm
        return comm

    @classmethod
    def close_all(cls):
        for c in cls._by_group.values():
            if c is not None:
                c.close()
        cls._by_group.clear()
        _OWN_GROUP.clear()          # (groups of a world that is about to be destroyed)


_OWN_GROUP = {}       # default-group object -> the process group SyncBN's torch.distributed collectives use (knob SYNCBN_OWN_GROUP)


def syncbn_collective_group(group):
    """-> the process group SyncBN's statistics exchanges run on.  For the DEFAULT group (group is None: what
    convert_sync_batchnorm(model) gives every layer, main.py:123) that is a group of the same ranks created once, collectively, at the
    first exchange -- every rank reaches it at the same program point, the first SyncBN layer of the first forward -- so that the
    exchanges get their own communicator and stream and never queue behind a gradient bucket's all-reduce on the default group's
    stream.  A caller-supplied subgroup is used as it is (dist.new_group would need the ranks outside it to take part)."""
    if group is not None:
        return group
    from . import tuning as _tuning
    knob = _tuning.host("SYNCBN_OWN_GROUP")
    if knob == 0:
        return group
    world = dist.get_world_size()
    if knob == 1 and (world < 2 or dist.get_backend() != "nccl"):
        return group
    key = dist.group.WORLD
    g = _OWN_GROUP.get(key)
    if g is None:
        g = _OWN_GROUP[key] = dist.new_group(ranks=list(range(world)), backend=dist.get_backend())
    return g


def sync_bn_forward(backend, x, weight, bias, residual, running_mean, running_var, nbt, eps, momentum, relu, group, conv_stats=None,
                    out_into=None):
    """SyncBN forward on the engine's split kernels: local (mean, M2, count) -> ONE all_gather of 2C+1 floats per rank -> Chan's
    parallel combination (+ running statistics) -> fused normalise (+residual) (+ReLU).  -> y, stats [2C], inv_n [1] (device)"""
    c = x.shape[1]
    world = dist.get_world_size(group)
    comm = EngineComm.get(group, x.device) if (x.is_cuda and conv_stats is None and hasattr(backend, "bn_forward_sync")) else None
    if comm is not None:        # ONE engine call: statistics -> ncclAllGather -> combine -> apply, all on the compute stream
        if _TIMING["on"]:
            _SYNCBN_EVENTS.append(None)
        return backend.bn_forward_sync(comm, x, weight, bias, eps, momentum, running_mean, running_var, nbt, residual, relu, out_into)
    # [mean | M2 | count], 2 kernels; the partial sums come from the producing conv's epilogue when it made them
    local = backend.bn_stats(x, conv_stats) if conv_stats is not None else backend.bn_stats(x)
    allst = torch.empty(world, 2 * c + 1, dtype=torch.float32, device=x.device)
    group = syncbn_collective_group(group)
    if dist.get_backend(group) == "nccl":
        _timed_collective(lambda: dist.all_gather_into_tensor(allst, local, group=group), x.is_cuda)
    else:  # gloo (single-GPU dry runs / CPU tests) has no flat all-gather
        _timed_collective(lambda: dist.all_gather(list(allst.unbind(0)), local, group=group), x.is_cuda)
    # one kernel: Chan's combination, running statistics, num_batches_tracked, 1/N (device scalar)
    stats, inv_n = backend.bn_sync_combine(allst, c, eps, momentum, running_mean, running_var, nbt)
    y = backend.bn_apply(x, weight, bias, stats, residual, relu, out_into) if out_into is not None else \
        backend.bn_apply(x, weight, bias, stats, residual, relu)
    return y, stats, inv_n


def sync_bn_backward(backend, x, y, dy, weight, bias, stats, inv_n, relu_mode, want_res, group, gparam, bparam):
    """SyncBN backward: local [sum dy' | sum dy' xhat] -> ONE all_reduce of 2C floats -> apply.  Parameter gradients stay local
    (DDP averages them): written by the reduce kernel, straight into the gradient-bucket slots when gparam / bparam own them.
    -> dx, dres, dgamma, dbeta, slots (True: dgamma / dbeta ARE the slot views)"""
    from .me.modules import grad_slot_view
    c = x.shape[1]
    gview = grad_slot_view(gparam) if gparam is not None else None
    bview = grad_slot_view(bparam) if bparam is not None else None
    if gview is None or bview is None:
        gview = bview = None
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    else:
        dgamma, dbeta = gview, bview
    comm = EngineComm.get(group, x.device) if (x.is_cuda and hasattr(backend, "bn_backward_sync")) else None
    if comm is not None:        # ONE engine call: reduce -> ncclAllReduce -> apply
        if _TIMING["on"]:
            _SYNCBN_EVENTS.append(None)
        dx, dres = backend.bn_backward_sync(comm, x, y, dy, weight, bias, stats, inv_n, relu_mode, want_res, dgamma, dbeta)
        return dx, dres, dgamma, dbeta, gview is not None
    sums = backend.bn_backward_reduce(x, y, dy, weight, bias, stats, relu_mode, dgamma, dbeta)
    group = syncbn_collective_group(group)
    _timed_collective(lambda: dist.all_reduce(sums, group=group), x.is_cuda)
    dx, dres = backend.bn_backward_apply(x, y, dy, weight, bias, stats, sums, inv_n, relu_mode, want_res)
    return dx, dres, dgamma, dbeta, gview is not None


class _SyncBNFused(torch.autograd.Function):
    """SyncBN as one autograd node around sync_bn_forward / sync_bn_backward (the whole-block node of models.py calls the
    same two functions)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, nbt, eps, momentum, relu, group, backend, conv_stats=None,
                out_into=None):
        # out_into (_CatSlot): y goes straight into a column slice of the concat buffer (zero-copy ME.cat; not a tensor input)
        y, stats, inv_n = sync_bn_forward(backend, x, weight, bias, residual, running_mean, running_var, nbt, eps, momentum, relu,
                                          group, conv_stats, (out_into.buf, out_into.off) if out_into is not None else None)
        ctx.backend, ctx.group, ctx.has_res = backend, group, residual is not None
        ctx.relu_mode = 0 if not relu else (1 if residual is not None else 2)
        ctx.gparam = weight if isinstance(weight, torch.nn.Parameter) else None
        ctx.bparam = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.save_for_backward(x, weight, bias, stats, y if ctx.relu_mode == 1 else x.new_empty(0), inv_n)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, stats, y, inv_n = ctx.saved_tensors
        yy = y if ctx.relu_mode == 1 else None      # (dy / y may be column slices of a concat buffer: the engine reads them in place)
        dx, dres, dgamma, dbeta, slots = sync_bn_backward(ctx.backend, x, yy, dy, weight, bias, stats, inv_n, ctx.relu_mode,
                                                          ctx.has_res and ctx.needs_input_grad[3], ctx.group, ctx.gparam, ctx.bparam)
        if slots:
            return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None
        return dx, dgamma.to(weight.dtype), dbeta.to(weight.dtype), dres, None, None, None, None, None, None, None, None, None, None


def sync_batch_norm(x, bn, group=None, residual=None, relu=False, conv_stats=None, out_into=None):
    """Batch statistics over the rows of all ranks.  Device tensors run on the engine's fused kernels; the pure
    torch formulation below is only reachable with CPU tensors (the gloo host-logic tests)."""
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    nbt = bn.num_batches_tracked if bn.track_running_stats else None
    if x.is_cuda:
        from .me.core import get_backend
        if conv_stats is not None and (conv_stats[1] is not None) != (rm is not None):
            conv_stats = None
        return _SyncBNFused.apply(x, bn.weight, bn.bias, residual, rm, rv, nbt, bn.eps, bn.momentum, relu, group, get_backend(), conv_stats,
                                  out_into)
    if nbt is not None:
        nbt += 1
    y = _SyncBNFunction.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, syncbn_collective_group(group))
    if residual is not None:
        y = y + residual
    if relu:
        y = torch.relu(y)
    return y
