"""Module surface: MinkowskiConvolution[Transpose], MinkowskiBatchNorm, MinkowskiSyncBatchNorm,
MinkowskiReLU, MinkowskiNetwork ... with the constructor signatures, parameter names and shapes the
reference relies on:
  kernel [K,Cin,Cout] ([Cin,Cout] for 1x1 stride 1), bias [1,Cout]     common.py:195-203,228-236
  MinkowskiBatchNorm.bn = nn.BatchNorm1d                               common.py:19, resnet.py:80-82
so released checkpoints' state-dict keys/shapes load (lib/utils.py:17-45).
"""
import math

import torch
import torch.nn as nn

from . import deferred as _deferred
from .core import SparseTensor, get_backend
from .kernel import KernelGenerator, RegionType, convert_to_int_list


from .. import tuning as _tuning
_WIDE_WGRAD_INLINE = _tuning.host("WIDE_WGRAD_INLINE") != 0
_DBG_WGRAD = _tuning.host("DBG_WGRAD")   # "", "skip", "inline": step-time attribution experiments only


class MinkowskiModuleBase(nn.Module):
    pass


_WGRAD_SEQ = [0]     # issue order of the side-stream weight gradients (the stream runs them in this order)


def _invalidate_packed():
    be = get_backend()
    if hasattr(be, "invalidate_packed_weights"):
        be.invalidate_packed_weights()


def _invalidate_packed_hook(module, incompatible_keys):
    """load_state_dict post-hook (a module-level function: hooks are pickled with the module)"""
    _invalidate_packed()


class MinkowskiNetwork(nn.Module):
    """models/model.py:4-16 subclasses this and stores D."""

    def __init__(self, D):
        super().__init__()
        self.D = D


# ------------------------------------------------------------------------------------------ convolution
def grad_slot_view(param):
    """Fresh view of the parameter's slot in a flat gradient bucket (languagegroundedsemseg_amd.ddp.BucketedDDP), or
    None.  Returning such a view from backward() lets autograd adopt it as `.grad` without an accumulation kernel:
    the engine has already written the gradient where the all-reduce and the optimiser will read it."""
    slot = getattr(param, "_lgs_grad_slot", None)
    if slot is None or param.grad is not None:
        return None
    flat, off, shape = slot
    n = 1
    for d in shape:
        n *= d
    return flat[off:off + n].view(shape)


def note_side_wgrad(kparam):
    """book-keeping of one weight gradient issued on the side stream (here, or by lgs_block_backward): its place in the stream's
    order and the step it belongs to -- BucketedDDP picks the newest event of a bucket from these"""
    _WGRAD_SEQ[0] += 1
    kparam._lgs_wgrad_seq = _WGRAD_SEQ[0]
    owner = getattr(kparam, "_lgs_ddp", None)
    kparam._lgs_wgrad_step = owner._step if owner is not None else -1


def conv_weight_grad(kmap, feats, gout, transposed, kparam, kshape, kdtype):
    """Weight gradient of one convolution.  When the kernel parameter owns a bucket slot the gradient is written straight
    into it ON A SIDE STREAM (off the backward critical path dgrad -> BN -> dgrad ...; both kernels are latency-bound, so
    running them concurrently fills the machine) and the slot view is returned for autograd to adopt as `.grad`."""
    view = grad_slot_view(kparam) if kparam is not None else None
    backend = get_backend()
    if view is None or not gout.is_cuda or not hasattr(backend, "side_stream"):
        return kmap.conv_wgrad(feats, gout, transposed).reshape(kshape).to(kdtype)
    if _DBG_WGRAD == "skip":
        return view                                      # profiling knob: no weight gradient at all
    # >= 256 x 256-channel layers: both the weight gradient (k_wgrad_wide / wide k_wgrad_ps) and the dgrad running beside it on the
    # compute stream (k_conv_wide) are MFMA-bound kernels that own their CUs -- concurrently they halve each other (round 3, CLIP
    # step: dgrad 17.5 ms in-step vs 9.2 alone, weight gradient 15.9 vs 7.7, and the 1x1 512 -> 544 dgrad next to them 11.6 vs
    # 1.3 ms), so there is nothing to overlap: they run back to back on the compute stream
    # (big maps only: the 256-channel layers of the coarse levels are latency-bound launches that DO overlap)
    wide = _WIDE_WGRAD_INLINE and feats.shape[1] >= 256 and gout.shape[1] >= 256 and gout.shape[0] >= 65536
    if _DBG_WGRAD == "inline" or wide or getattr(kmap.mgr, "inline_wgrad", False):
        # small (host-bound) batches, wide layers, or the profiling knob: weight gradient on the compute stream
        kmap.conv_wgrad(feats, gout, transposed, out=view.view(kmap.K, -1, kshape[-1]))
        return view
    dev = gout.device
    side = backend.side_stream(dev)
    fork = backend.fork_event(dev)
    fork.record(backend.current_stream(dev))             # feats / gout are ready
    side.wait_event(fork)
    kmap.conv_wgrad(feats, gout, transposed, out=view.view(kmap.K, -1, kshape[-1]), stream=side)
    # one reusable event per parameter: BucketedDDP waits for exactly the weight gradients of the bucket it is about to
    # reduce (ddp._wait_bucket_wgrads), not for the whole side stream
    ev = getattr(kparam, "_lgs_wgrad_event", None)
    if ev is None:
        ev = kparam._lgs_wgrad_event = torch.cuda.Event()
    ev.record(side)
    note_side_wgrad(kparam)
    feats.record_stream(side)
    gout.record_stream(side)
    return view                                          # consumers wait for the side stream in BucketedDDP


def _bias_grad(gout):
    """column sums of the output gradient in fp32.  On the engine: the BatchNorm statistics reduction (two launches at the
    streaming rate: per-block sums about a pivot, folded in double in a fixed order) read as mean x count -- torch's reduction
    of the [1.2 M, 200] bf16 gradient of the classifier took 190 us of the step."""
    backend = get_backend()
    c = gout.shape[1]
    if hasattr(backend, "bn_stats") and gout.is_cuda and gout.is_contiguous() and gout.shape[0] > 0 and \
            c % (8 if gout.dtype == torch.bfloat16 else 4) == 0 and gout.dtype in (torch.bfloat16, torch.float32):
        rec = backend.bn_stats(gout)                           # [mean | M2 | count]
        return (rec[:c] * rec[2 * c]).view(1, c)
    return gout.sum(0, keepdim=True, dtype=torch.float32)      # fp32 accumulation without an fp32 copy of gout


class MinkowskiConvolutionFunction(torch.autograd.Function):
    """out = conv(in) over a cached kernel map; backward = dgrad + wgrad (conv_weight_grad) (+ bias grad) on the same map."""

    @staticmethod
    def forward(ctx, feats, kernel, bias, kmap, transposed, bn_pivot=None, want_bn_stats=False, pack_cache=None):
        ctx.kmap, ctx.transposed, ctx.has_bias = kmap, transposed, bias is not None
        ctx.pack_cache = pack_cache
        kw = {"pack_cache": pack_cache} if pack_cache is not None else {}
        ctx.kshape = kernel.shape
        ctx.kparam = kernel if isinstance(kernel, torch.nn.Parameter) else None
        ctx.save_for_backward(feats, kernel)
        if want_bn_stats:
            # the conv epilogue also emits per-tile sum / sum-of-squares of its output for the BatchNorm that follows
            out, stats = kmap.conv_forward(feats, kernel, bias, transposed, bn_pivot=bn_pivot, want_bn_stats=True, **kw)
            ctx.bn_stats = stats                 # picked up by MinkowskiConvolutionBase.forward (not a graph output)
            return out
        return kmap.conv_forward(feats, kernel, bias, transposed, **kw)

    @staticmethod
    def backward(ctx, gout):
        feats, kernel = ctx.saved_tensors
        gout = gout.contiguous()
        gin = gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = conv_weight_grad(ctx.kmap, feats, gout, ctx.transposed, ctx.kparam, ctx.kshape, kernel.dtype)
        if ctx.needs_input_grad[0]:
            if getattr(ctx, "pack_cache", None) is not None:
                gin = ctx.kmap.conv_dgrad(gout, kernel, ctx.transposed, pack_cache=ctx.pack_cache)
            else:
                gin = ctx.kmap.conv_dgrad(gout, kernel, ctx.transposed)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _bias_grad(gout)
        return gin, gw, gb, None, None, None, None, None


MinkowskiConvolutionTransposeFunction = MinkowskiConvolutionFunction


class _StatsHolder:
    stats = None


class _ConvStatsFunction(MinkowskiConvolutionFunction):
    """MinkowskiConvolutionFunction whose forward also asks the kernel for the next norm's statistics (side output)."""

    @staticmethod
    def forward(ctx, feats, kernel, kmap, transposed, bn_pivot, holder, pack_cache=None):
        out = MinkowskiConvolutionFunction.forward(ctx, feats, kernel, None, kmap, transposed, bn_pivot, True, pack_cache)
        holder.stats = ctx.bn_stats
        ctx.bn_stats = None
        return out

    @staticmethod
    def backward(ctx, gout):
        gin, gw = MinkowskiConvolutionFunction.backward(ctx, gout)[:2]
        return gin, gw, None, None, None, None, None


def _conv_with_stats(feats, kernel, kmap, transposed, pivot, holder, pack_cache=None):
    return _ConvStatsFunction.apply(feats, kernel, kmap, transposed, pivot, holder, pack_cache)


class MinkowskiConvolutionBase(MinkowskiModuleBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 is_transpose=False, expand_coordinates=False, convolution_mode=None, dimension=-1):
        super().__init__()
        assert dimension > 0, "dimension must be a positive integer"
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               expand_coordinates=expand_coordinates, dimension=dimension)
        self.is_transpose = is_transpose
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_generator = kernel_generator
        self.dimension = dimension
        self.use_mm = False
        self.kernel_size = kernel_generator.kernel_size
        self.stride = kernel_generator.kernel_stride
        self.dilation = kernel_generator.kernel_dilation
        if kernel_generator.region_type != RegionType.HYPER_CUBE:
            raise NotImplementedError("only HYPER_CUBE regions are part of the D=3 model family")
        if any(d != 1 for d in self.dilation):
            raise NotImplementedError("dilation != 1 is not used by the model family")
        if len(set(self.kernel_size)) != 1 or len(set(self.stride)) != 1:
            raise NotImplementedError("anisotropic kernels/strides are not used on D=3")
        ks, st = self.kernel_size[0], self.stride[0]
        if not ((ks == 3 and st == 1) or (ks == 2 and st == 2) or (ks == 1 and st == 1)):
            raise NotImplementedError("kernel_size/stride combination (%d,%d) is not part of the model family" % (ks, st))
        self.kernel_volume = kernel_generator.kernel_volume
        if self.kernel_volume == 1 and st == 1:
            self.use_mm = True
            kshape = (in_channels, out_channels)
        else:
            kshape = (self.kernel_volume, in_channels, out_channels)
        self._pack_cache = {}        # packed weight images of this module's launch shapes (backend_hip.PackedWeights)
        self.kernel = nn.Parameter(torch.empty(kshape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty((1, out_channels), dtype=torch.float32)) if bias else None
        self.reset_parameters()
        # load_state_dict() copies into `.data` without bumping the parameter's version counter: drop the packed images
        self.register_load_state_dict_post_hook(_invalidate_packed_hook)

    def __getstate__(self):
        # packed weight images are a device-side cache tied to this process: never pickled / deep-copied with the module
        state = self.__dict__.copy()
        state["_pack_cache"] = {}
        return state

    def reset_parameters(self, is_transpose=None):
        is_transpose = self.is_transpose if is_transpose is None else is_transpose
        with torch.no_grad():
            n = (self.out_channels if is_transpose else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.data.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.data.uniform_(-stdv, stdv)
        _invalidate_packed()      # `.data` writes are invisible to the version counter the packed-image cache keys on

    def forward(self, input, coordinates=None, bn=None):
        """Standard call: `conv(x)` records the convolution (me/deferred.py) and returns a tensor whose features are pending; the
        coordinate map / kernel map it needs are requested now (built on the engine's map stream).
        `bn` (extension, used by the executor and by tests): the MinkowskiBatchNorm this output goes to next; runs immediately."""
        assert isinstance(input, SparseTensor)
        nch = input._nch()
        assert nch == self.in_channels, "Channel size mismatch %d != %d" % (nch, self.in_channels)
        if bn is None and _deferred.ENABLED:
            return _deferred.record_conv(self, input, self._resolve(input))
        return self._forward_now(input, bn)

    def _resolve(self, input):
        """-> (output coordinate map key, kernel map, transposed)"""
        mgr = input.coordinate_manager
        ks, st = self.kernel_size[0], self.stride[0]
        in_key = input.coordinate_map_key
        if not self.is_transpose:
            out_key = in_key if st == 1 else mgr.stride(in_key, st)
            return out_key, mgr.kernel_map_handle(in_key, out_key, ks), False
        out_key = in_key if st == 1 else mgr.finer_key(in_key)
        # the transposed conv reuses the forward map of the matching strided conv, in/out swapped
        return out_key, mgr.kernel_map_handle(out_key, in_key, ks), True

    def _forward_now(self, input, bn=None, resolved=None, out=None):
        """run the convolution.  `bn`: in training the conv epilogue may also produce that norm's batch statistics (pivoted on its
        running mean; off by default, CONV_BN_STATS) and hand them over on the output tensor.  `out`: the pending tensor to fill."""
        mgr = input.coordinate_manager
        out_key, kmap, transposed = resolved if resolved is not None else self._resolve(input)
        x = input.F
        stats = None
        want = (bn is not None and getattr(get_backend(), "conv_bn_stats", False) and bn.bn.training and bn.bn.affine and x.is_cuda
                and self.bias is None
                and get_backend().want_conv_bn_stats(mgr.size(out_key), self.out_channels, x.element_size()))
        if want:
            pivot = bn.bn.running_mean if bn.bn.track_running_stats else None
            holder = _StatsHolder()
            y = _conv_with_stats(x, self.kernel, kmap, transposed, pivot, holder, self._cache_for(x))
            stats = holder.stats                   # (partials, pivot) or None
        else:
            y = MinkowskiConvolutionFunction.apply(x, self.kernel, self.bias, kmap, transposed, None, False, self._cache_for(x))
        if out is None:
            out = SparseTensor(y, coordinate_map_key=out_key, coordinate_manager=mgr)
        else:
            out._F, out._op = y, None
        out._bn_stats = stats
        return out

    def _cache_for(self, feats):
        """packed-image cache of this module, only on the HIP backend with an fp32 master weight"""
        if not feats.is_cuda or not getattr(get_backend(), "weights_updated", None) or self.kernel.dtype != torch.float32:
            return None
        return self._pack_cache

    def __repr__(self):
        return "%s(in=%d, out=%d, kernel_size=%s, stride=%s, dilation=%s)" % (
            self.__class__.__name__, self.in_channels, self.out_channels, self.kernel_size, self.stride, self.dilation)


class MinkowskiConvolution(MinkowskiConvolutionBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=False, expand_coordinates=expand_coordinates, convolution_mode=convolution_mode,
                         dimension=dimension)


class MinkowskiConvolutionTranspose(MinkowskiConvolutionBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=True, expand_coordinates=expand_coordinates, convolution_mode=convolution_mode,
                         dimension=dimension)


# ------------------------------------------------------------------------------------------ normalisation
class FusedBatchNormFunction(torch.autograd.Function):
    """y = relu?(BN(x) (+ residual)) with batch statistics, one engine call each way."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, relu, backend, nbt=None, conv_stats=None,
                out_into=None):
        ctx.gparam = gamma if isinstance(gamma, torch.nn.Parameter) else None
        ctx.bparam = beta if isinstance(beta, torch.nn.Parameter) else None
        if out_into is not None:       # _CatSlot: y goes straight into a column slice of the concat buffer (not a tensor input)
            y, stats = backend.bn_forward(x, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, nbt,
                                          conv_stats=conv_stats, out_into=(out_into.buf, out_into.off))
        elif conv_stats is not None:
            y, stats = backend.bn_forward(x, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, nbt,
                                          conv_stats=conv_stats)
        elif nbt is not None:
            y, stats = backend.bn_forward(x, gamma, beta, eps, momentum, running_mean, running_var, residual, relu, nbt)
        else:
            y, stats = backend.bn_forward(x, gamma, beta, eps, momentum, running_mean, running_var, residual, relu)
        # ReLU mask in the backward: recomputed from x when there is no residual (mode 2, y is not kept alive),
        # taken from the saved output otherwise (mode 1)
        ctx.backend, ctx.has_res = backend, residual is not None
        ctx.relu_mode = 0 if not relu else (1 if residual is not None else 2)
        if ctx.relu_mode == 1:
            ctx.save_for_backward(x, gamma, beta, stats, y)
        else:
            ctx.save_for_backward(x, gamma, beta, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        x, gamma, beta, stats = saved[0], saved[1], saved[2], saved[3]
        y = saved[4] if ctx.relu_mode == 1 else None
        gview = grad_slot_view(ctx.gparam) if ctx.gparam is not None else None
        bview = grad_slot_view(ctx.bparam) if ctx.bparam is not None else None
        if gview is None or bview is None or not hasattr(ctx.backend, "side_stream"):
            gview = bview = None
        dx, dres, dgamma, dbeta = ctx.backend.bn_backward(x, y, dy, gamma, beta, stats, ctx.relu_mode,
                                                          ctx.has_res and ctx.needs_input_grad[3], gview, bview)
        if gview is not None:
            return dx, gview, bview, dres, None, None, None, None, None, None, None, None, None
        return dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), dres, None, None, None, None, None, None, None, None, None


class _CatSlot:
    """where a norm writes its output for a zero-copy ME.cat: column offset `off` of the [N, C_total] concat buffer `buf`
    (a plain Python object on purpose: autograd must not see the buffer as a tensor input of the norm)"""

    partner = None       # the skip tensor whose features were COPIED into the right-hand columns (half-copy concat)

    def __init__(self, buf, off, width):
        self.buf, self.off, self.width = buf, off, width


class EvalBatchNormFunction(torch.autograd.Function):
    """eval-mode BatchNorm (running statistics) (+ residual) (+ ReLU) on the engine's apply kernels: forward =
    lgs_bn_apply with stats = [running_mean | 1/sqrt(running_var + eps)]; backward (frozen statistics: fine-tuning with
    BN in eval mode) = lgs_bn_backward_reduce for d gamma / d beta and lgs_bn_backward_apply with zero batch sums."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, stats, relu, backend):
        y = backend.bn_apply(x, gamma, beta, stats, residual, relu)
        ctx.backend, ctx.has_res = backend, residual is not None
        ctx.relu_mode = 0 if not relu else (1 if residual is not None else 2)
        ctx.save_for_backward(x, gamma, beta, stats, y if ctx.relu_mode == 1 else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats, y = ctx.saved_tensors
        dy = dy.contiguous()
        yy = y if ctx.relu_mode == 1 else None
        c = x.shape[1]
        sums = ctx.backend.bn_backward_reduce(x, yy, dy, gamma, beta, stats, ctx.relu_mode)
        zero = torch.zeros_like(sums)
        dx, dres = ctx.backend.bn_backward_apply(x, yy, dy, gamma, beta, stats, zero, 0.0, ctx.relu_mode,
                                                 ctx.has_res and ctx.needs_input_grad[3])
        return dx, sums[c:].to(gamma.dtype), sums[:c].to(gamma.dtype), dres, None, None, None


class MinkowskiBatchNorm(nn.Module):
    """`.bn` is a plain nn.BatchNorm1d so state-dict keys are `*.bn.{weight,bias,running_*}`."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, input, relu=False, residual=None, cat_up=0, cat_into=None):
        """Standard call: `norm(x)` records the normalisation (me/deferred.py); a following MinkowskiReLU(inplace=True) and
        `out += residual` become its epilogue, a later `me.cat` its output placement, and the whole thing runs as one kernel when
        a value is first needed.
        `relu` / `residual` / `cat_up` / `cat_into` are the executor's (and the per-op tests') way of asking for that fused kernel
        directly; with any of them the call runs immediately.  Zero-copy ME.cat(up, skip) (res16unet.py:237,247,257,267): the norm
        that produces the SKIP tensor gets cat_up = channels of the future `up` half: it allocates the [N, cat_up + C] concat buffer
        and writes its output into the right-hand columns; the norm that produces `up` gets cat_into = that skip tensor and writes
        into the left-hand columns; ME.cat then returns the buffer itself.  Both are hints: paths that cannot honour them (eval
        mode, CPU oracle backend) return ordinary tensors and ME.cat copies."""
        if _deferred.ENABLED and not relu and residual is None and not cat_up and cat_into is None:
            return _deferred.record_bn(self, input)
        return self._forward_now(input, relu, residual, cat_up, cat_into)

    def train(self, mode=True):
        if mode != self.training:
            _deferred.flush_all()          # recorded calls run with the mode they were recorded in
        return super().train(mode)

    def _forward_now(self, input, relu=False, residual=None, cat_up=0, cat_into=None, out=None):
        y, slot = self._run(input, relu, residual, cat_up, cat_into)
        if out is None:
            out = SparseTensor(y, coordinate_map_key=input.coordinate_map_key, coordinate_manager=input.coordinate_manager)
        else:
            out._F, out._op = y, None
        out._cat_slot = slot
        return out

    def _run(self, input, relu, residual, cat_up, cat_into):
        """-> (features, _CatSlot | None)"""
        bn = self.bn
        backend = get_backend()
        x = input.F
        res = residual.F if isinstance(residual, SparseTensor) else residual
        fused = hasattr(backend, "bn_forward") and bn.affine and (bn.training or not bn.track_running_stats)
        if fused:
            rm = bn.running_mean if bn.track_running_stats else None
            rv = bn.running_var if bn.track_running_stats else None
            nbt = bn.num_batches_tracked if bn.track_running_stats else None
            if nbt is not None and not getattr(backend, "bn_counts_batches", False):
                nbt += 1
                nbt = None                                 # (the HIP engine increments it inside the fold kernel)
            cs = input._bn_stats
            if cs is not None and (cs[1] is not None) != (rm is not None):
                cs = None                                  # pivot convention mismatch (cannot happen for the conv's own bn)
            slot = self._cat_slot_for(x, cat_up, cat_into, backend) if type(self) is MinkowskiBatchNorm else None
            y = FusedBatchNormFunction.apply(x, bn.weight, bn.bias, res, rm, rv, bn.eps, bn.momentum, relu, backend, nbt, cs, slot)
            return y, slot
        elif hasattr(backend, "bn_apply") and bn.affine and bn.track_running_stats and x.is_cuda:
            # eval mode on the engine too (inference / validation passes, BN frozen during fine-tuning)
            # [running_mean | 1/sqrt(running_var + eps)], rebuilt only when the running statistics were written (a frozen trunk
            # otherwise launches three tiny torch kernels per norm and step: 186 launches for Res16UNet34C)
            key = (bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps)
            cached = getattr(self, "_eval_stats", None)
            if cached is None or cached[0] != key or cached[1].device != x.device:
                with torch.no_grad():
                    stats = torch.cat([bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)])
                self._eval_stats = cached = (key, stats)
            stats = cached[1]
            y = EvalBatchNormFunction.apply(x, bn.weight, bn.bias, res, stats, relu, backend)
        else:   # CPU oracle backend of the tests / norms without affine parameters
            y = bn(x.float()).to(x.dtype) if x.dtype != torch.float32 else bn(x)
            if res is not None:
                y = y + res
            if relu:
                y = torch.relu(y)
        return y, None

    @staticmethod
    def _cat_slot_for(x, cat_up, cat_into, backend):
        """the column slice of a concat buffer this norm's output goes to (zero-copy ME.cat), or None"""
        if not (getattr(backend, "bn_out_into", False) and x.is_cuda):
            return None
        c = x.shape[1]
        al = 8 if x.dtype == torch.bfloat16 else 4
        if cat_up > 0 and cat_up % al == 0 and c % al == 0:
            buf = torch.empty((x.shape[0], cat_up + c), dtype=x.dtype, device=x.device)
            return _CatSlot(buf, cat_up, c)
        if cat_into is not None:
            other = cat_into._cat_slot
            if other is not None:
                if (other.off == c and other.buf.shape[0] == x.shape[0] and other.buf.dtype == x.dtype
                        and not getattr(other, "taken", False)):
                    other.taken = True
                    return _CatSlot(other.buf, 0, c)
                return None
            # the skip tensor exists already as an ordinary tensor: this norm writes the left-hand columns of a fresh concat
            # buffer, the skip half is copied in once (me/deferred.py _cat_partner)
            if cat_into._op is None and c % al == 0:
                sf = cat_into._F
                if (sf.is_cuda and sf.dim() == 2 and sf.shape[0] == x.shape[0] and sf.dtype == x.dtype and sf.shape[1] % al == 0):
                    buf = torch.empty((x.shape[0], c + sf.shape[1]), dtype=x.dtype, device=x.device)
                    with torch.no_grad():
                        buf[:, c:].copy_(sf)
                    slot = _CatSlot(buf, 0, c)
                    slot.partner = cat_into
                    return slot
        return None

    def __repr__(self):
        b = self.bn
        return "MinkowskiBatchNorm(%d, eps=%g, momentum=%g, affine=%s, track_running_stats=%s)" % (
            b.num_features, b.eps, b.momentum, b.affine, b.track_running_stats)


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """Cross-rank batch statistics (main.py:122-123) on the engine's split BN kernels (ddp._SyncBNFused): every rank
    contributes one [mean | M2 | count] record to ONE all-gather per layer, the records are combined with Chan's
    parallel formula in one kernel; backward exchanges [sum dy | sum dy xhat] with ONE all-reduce of 2C floats.
    `force_sync` (class attribute, tests only) takes the collective path even in a world of one rank."""
    force_sync = False

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group

    def _run(self, input, relu, residual, cat_up, cat_into):
        import torch.distributed as dist
        if not (self.training and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size(self.process_group) > 1 or MinkowskiSyncBatchNorm.force_sync)):
            return super()._run(input, relu, residual, cat_up, cat_into)
        from ..ddp import sync_batch_norm
        res = residual.F if isinstance(residual, SparseTensor) else residual
        backend = get_backend()
        x = input.F
        slot = self._cat_slot_for(x, cat_up, cat_into, backend) if hasattr(backend, "bn_forward_sync") else None
        y = sync_batch_norm(x, self.bn, self.process_group, residual=res, relu=relu, conv_stats=input._bn_stats, out_into=slot)
        return y, slot

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        out = module
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            b = module.bn
            out = cls(b.num_features, b.eps, b.momentum, b.affine, b.track_running_stats, process_group)
            if b.affine:
                with torch.no_grad():
                    out.bn.weight = b.weight
                    out.bn.bias = b.bias
            out.bn.running_mean = b.running_mean
            out.bn.running_var = b.running_var
            out.bn.num_batches_tracked = b.num_batches_tracked
            out.train(module.training)
        for name, child in module.named_children():
            out.add_module(name, cls.convert_sync_batchnorm(child, process_group))
        return out


class MinkowskiInstanceNorm(nn.Module):
    """Per-scene normalisation (only the out-of-scope 34Dv2/v3 variants use it, clip_models.py:416,431)."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.eps = 1e-6
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, input):
        x, b = input.F, input.C[:, 0].long()
        nb = int(b.max().item()) + 1 if b.numel() else 0
        cnt = torch.zeros(nb, device=x.device, dtype=torch.float32).index_add_(0, b, torch.ones_like(b, dtype=torch.float32))
        xf = x.float()
        mean = torch.zeros(nb, x.shape[1], device=x.device).index_add_(0, b, xf) / cnt[:, None]
        d = xf - mean[b]
        var = torch.zeros(nb, x.shape[1], device=x.device).index_add_(0, b, d * d) / cnt[:, None]
        y = d / torch.sqrt(var[b] + self.eps) * self.weight + self.bias
        return input._like(y.to(x.dtype))


# ------------------------------------------------------------------------------------------ pointwise
class MinkowskiNonlinearityBase(MinkowskiModuleBase):
    MODULE = None

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.module = self.MODULE(*args, **kwargs)

    def forward(self, input):
        return SparseTensor(self.module(input.F), coordinate_map_key=input.coordinate_map_key,
                            coordinate_manager=input.coordinate_manager)

    def __repr__(self):
        return self.__class__.__name__ + "()"


class MinkowskiReLU(MinkowskiNonlinearityBase):
    MODULE = nn.ReLU

    def forward(self, input):
        # in place on the pending result of a norm nobody has read yet: that norm's ReLU epilogue (me/deferred.py); the SAME tensor
        # object comes back -- ME returns a new wrapper of the same (in-place rectified) storage, the two are indistinguishable
        if input._op is not None and self.module.inplace and _deferred.relu_inplace(input):
            return input
        return super().forward(input)


class MinkowskiSigmoid(MinkowskiNonlinearityBase):
    MODULE = nn.Sigmoid


class MinkowskiTanh(MinkowskiNonlinearityBase):
    MODULE = nn.Tanh


class MinkowskiLeakyReLU(MinkowskiNonlinearityBase):
    MODULE = nn.LeakyReLU


class MinkowskiELU(MinkowskiNonlinearityBase):
    MODULE = nn.ELU


class MinkowskiDropout(MinkowskiNonlinearityBase):
    MODULE = nn.Dropout


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, input):
        return input._like(self.linear(input.F))


class _OutOfScope(nn.Module):
    """Placeholder for ME modules that only out-of-scope model families construct:
    constructible (so `import models` and unrelated ctors work) but raises if executed."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError("%s is outside the Res16UNet hot path (SURVEY.md section 8)" % self.__class__.__name__)


class MinkowskiSumPooling(_OutOfScope):
    pass


class MinkowskiAvgPooling(_OutOfScope):
    pass


class MinkowskiMaxPooling(_OutOfScope):
    pass


class MinkowskiAvgUnpooling(_OutOfScope):
    pass


class MinkowskiPoolingTranspose(_OutOfScope):
    pass


class MinkowskiGlobalPooling(_OutOfScope):
    pass


class MinkowskiGlobalSumPooling(_OutOfScope):
    pass


class MinkowskiGlobalAvgPooling(_OutOfScope):
    pass


class MinkowskiGlobalMaxPooling(_OutOfScope):
    pass


class MinkowskiBroadcast(_OutOfScope):
    pass


class MinkowskiBroadcastAddition(_OutOfScope):
    pass


class MinkowskiBroadcastMultiplication(_OutOfScope):
    pass


class MinkowskiBroadcastConcatenation(_OutOfScope):
    pass
