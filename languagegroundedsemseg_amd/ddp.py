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


class BucketedDDP:
    """force_collectives=True issues the all-reduces even with a world of one rank (used by the single-GPU RCCL test:
    the collective path, its stream ordering and the bucket views are then the ones an 8-GPU run executes)."""

    def __init__(self, module, bucket_mb=32.0, process_group=None, force_collectives=False):
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
        if self.reduce:
            for p in params:
                dist.broadcast(p.data, src=0, group=self.group)
            for b in module.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b.data, src=0, group=self.group)

    def _make_bucket(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        off = 0
        bucket = {"flat": flat, "params": params, "pending": len(params), "n": len(params), "views": [], "launched": False}
        for p in params:
            # the engine's backward kernels write gradients straight into this slot (me.modules.grad_slot_view);
            # any other producer falls back to autograd's own accumulation into the same memory
            p._lgs_grad_slot = (flat, off, tuple(p.shape))
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
                self._launch(bucket)
            elif bucket["pending"] < 0:
                raise RuntimeError("BucketedDDP: a gradient arrived after its bucket was reduced -- call zero_grad() before "
                                   "every backward, or wrap all but the last micro-batch of an accumulation in no_sync()")
        return fn

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

    def _launch(self, bucket):
        self._join_side()
        self._collect(bucket)
        bucket["launched"] = True
        if self.world > 1:
            bucket["flat"].div_(self.world)
        self._handles.append(dist.all_reduce(bucket["flat"], group=self.group, async_op=True))

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()
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

    def no_sync(self):
        """Gradient accumulation (the reference's insseg trainer supports iter_size > 1): backward passes inside the
        context only accumulate into the flat buckets; the first backward outside it reduces them."""
        return BucketedDDP._NoSync(self)

    def finalize(self):
        """call after backward(): waits for the side-stream weight gradients and the in-flight bucket all-reduces
        (and flushes buckets whose parameters received no gradient this step -- the reference runs
        find_unused_parameters=True).  Buckets that were not launched from a hook are flushed in fixed bucket order, the
        same on every rank.  Inside no_sync() nothing is reduced."""
        self._join_side()
        if self.reduce and not self._defer:
            for b in self.buckets:
                if not b["launched"]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
            for b in self.buckets:      # ready for the next backward even if the caller clears grads some other way
                b["pending"] = b["n"]
                b["launched"] = False
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
            flat_p = torch.empty_like(b["flat"])
            for p, off in b["views"]:
                flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + p.numel()].view_as(p)
            self.state.append({"p": flat_p, "buf": None, "mask": torch.ones_like(flat_p)})
        self.steps = 0

    @torch.no_grad()
    def step(self):
        fused = self.state and self.state[0]["p"].is_cuda
        for b, st in zip(self.ddp.buckets, self.state):
            g = b["flat"]
            # parameters without a gradient this step (grad is None) must not move, not even by weight decay / momentum
            unused = [(p, off) for p, off in b["views"] if p.grad is None]
            if unused:
                st["mask"].fill_(1.0)
                for p, off in unused:
                    st["mask"][off:off + p.numel()] = 0.0
            if fused:   # one kernel per bucket (lgs_sgd_step) instead of four elementwise passes
                import ctypes
                from . import engine
                first = st["buf"] is None
                if first and self.momentum != 0:
                    st["buf"] = torch.empty_like(g)
                vp = ctypes.c_void_p
                with torch.cuda.device(g.device):
                    engine.check(engine.lib().lgs_sgd_step(
                        vp(st["p"].data_ptr()), vp(g.data_ptr()), vp(st["buf"].data_ptr()) if st["buf"] is not None else vp(None),
                        vp(st["mask"].data_ptr()) if unused else vp(None), int(g.numel()), float(self.lr), float(self.momentum),
                        float(self.dampening), float(self.weight_decay), int(first), vp(torch.cuda.current_stream(g.device).cuda_stream)))
                continue
            d = g.add(st["p"], alpha=self.weight_decay) if self.weight_decay != 0 else g.clone()
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


class _SyncBNFused(torch.autograd.Function):
    """SyncBN on the engine's split kernels (lgs_bn_stats / lgs_bn_apply / lgs_bn_backward_reduce /
    lgs_bn_backward_apply): local (mean, M2, count) -> ONE all_gather of 2C+1 floats per rank -> Chan's parallel
    combination -> fused normalise(+residual)(+ReLU); backward: local sums -> ONE all_reduce of 2C floats."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, nbt, eps, momentum, relu, group, backend, conv_stats=None):
        c = x.shape[1]
        world = dist.get_world_size(group)
        # [mean | M2 | count], 2 kernels; the partial sums come from the producing conv's epilogue when it made them
        local = backend.bn_stats(x, conv_stats) if conv_stats is not None else backend.bn_stats(x)
        allst = torch.empty(world, 2 * c + 1, dtype=torch.float32, device=x.device)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(allst, local, group=group)
        else:  # gloo (single-GPU dry runs / CPU tests) has no flat all-gather
            dist.all_gather(list(allst.unbind(0)), local, group=group)
        # one kernel: Chan's combination, running statistics, num_batches_tracked, 1/N (device scalar)
        stats, inv_n = backend.bn_sync_combine(allst, c, eps, momentum, running_mean, running_var, nbt)
        y = backend.bn_apply(x, weight, bias, stats, residual, relu)
        ctx.backend, ctx.group, ctx.has_res = backend, group, residual is not None
        ctx.relu_mode = 0 if not relu else (1 if residual is not None else 2)
        ctx.gparam = weight if isinstance(weight, torch.nn.Parameter) else None
        ctx.bparam = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.save_for_backward(x, weight, bias, stats, y if ctx.relu_mode == 1 else x.new_empty(0), inv_n)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .me.modules import grad_slot_view
        x, weight, bias, stats, y, inv_n = ctx.saved_tensors
        dy = dy.contiguous()
        yy = y if ctx.relu_mode == 1 else None
        c = x.shape[1]
        # parameter gradients stay local (DDP averages them): written by the reduce kernel, straight into the gradient
        # bucket slots when the parameters have them
        gview = grad_slot_view(ctx.gparam) if ctx.gparam is not None else None
        bview = grad_slot_view(ctx.bparam) if ctx.bparam is not None else None
        if gview is None or bview is None:
            gview = bview = None
            dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        else:
            dgamma, dbeta = gview, bview
        sums = ctx.backend.bn_backward_reduce(x, yy, dy, weight, bias, stats, ctx.relu_mode, dgamma, dbeta)
        dist.all_reduce(sums, group=ctx.group)
        dx, dres = ctx.backend.bn_backward_apply(x, yy, dy, weight, bias, stats, sums, inv_n, ctx.relu_mode,
                                                 ctx.has_res and ctx.needs_input_grad[3])
        if gview is not None:
            return dx, gview, bview, dres, None, None, None, None, None, None, None, None, None
        return dx, dgamma.to(weight.dtype), dbeta.to(weight.dtype), dres, None, None, None, None, None, None, None, None, None


def sync_batch_norm(x, bn, group=None, residual=None, relu=False, conv_stats=None):
    """Batch statistics over the rows of all ranks.  Device tensors run on the engine's fused kernels; the pure
    torch formulation below is only reachable with CPU tensors (the gloo host-logic tests)."""
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    nbt = bn.num_batches_tracked if bn.track_running_stats else None
    if x.is_cuda:
        from .me.core import get_backend
        if conv_stats is not None and (conv_stats[1] is not None) != (rm is not None):
            conv_stats = None
        return _SyncBNFused.apply(x, bn.weight, bn.bias, residual, rm, rv, nbt, bn.eps, bn.momentum, relu, group, get_backend(), conv_stats)
    if nbt is not None:
        nbt += 1
    y = _SyncBNFunction.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, group)
    if residual is not None:
        y = y + residual
    if relu:
        y = torch.relu(y)
    return y
