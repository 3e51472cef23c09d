"""One autograd node (and one engine call per direction) for a residual block.

The deferred executor (me/deferred.py) hands a run of recorded calls that IS a BasicBlock of the reference
(/root/reference/models/modules/resnet_block.py:41-57, NoReluBlock :133-161: conv3 -> norm -> relu -> conv3 -> norm
[-> 1x1 conv -> norm on the input] -> += residual [-> relu]) to `try_block`; the modules are whatever the caller built -- the
reference's own BasicBlock, the build's models.BasicBlock, or loose calls -- only the recorded dataflow is matched."""
import torch
import torch.nn as nn

from .. import tuning as _tuning
from .core import get_backend
from . import modules as _mod
from . import deferred as _d

# ---------------------------------------------------------------------------------------------- whole-block autograd node
# Op by op a BasicBlock is 4 (6 with a downsample branch) autograd nodes, as many module calls and SparseTensor wrappers each
# way; at one scene per step (~150 k voxels) the training step is bound by exactly that host work.
# The fast path issues the same engine calls, in the same order, with the same arguments, from ONE autograd node:
#   forward   conv1 -> norm1+ReLU -> conv2 -> [downsample conv 1x1
#             -> its norm] -> norm2 + residual (+ ReLU)
#   backward  norm2 -> {wgrad2 on the side stream, dgrad2} -> norm1 -> {wgrad1, dgrad1} [-> downsample norm -> {wgrad, dgrad}]
#             with the residual branch's gradient added in dgrad1's epilogue (lgs_conv_dgrad_accumulate: autograd's
#             accumulation of the two branches, same rounding, without the elementwise pass)
# It is taken only when every module of the block is a plain training-mode MinkowskiConvolution / MinkowskiBatchNorm on the
# HIP backend with no hooks attached (the layer-wise parity tests hook the modules and so run the op-by-op path);
# LGS_BLOCK_FUSED=0 turns it off.
_BLOCK_FUSED = _tuning.host("BLOCK_FUSED") != 0


def _plain(m):
    return not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


def _block_fast_path_ok(blk, xf):
    """blk: a BlockView; xf: the block's input features"""
    if not _BLOCK_FUSED or blk.cat_up or not xf.is_cuda or not torch.is_grad_enabled():
        return False
    from torch.nn.modules import module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks:
        return False
    be = get_backend()
    if not (hasattr(be, "bn_forward") and hasattr(be, "side_stream") and getattr(be, "bn_counts_batches", False)):
        return False
    convs, norms = [blk.conv1, blk.conv2], [blk.norm1, blk.norm2]
    if blk.downsample is not None:
        if len(blk.downsample) != 2:
            return False
        convs.append(blk.downsample[0]); norms.append(blk.downsample[1])
    for i, c in enumerate(convs):
        if type(c) is not _mod.MinkowskiConvolution or c.bias is not None or c.kernel.dtype != torch.float32 or not _plain(c):
            return False
        # the fast path hard-codes the kernel maps (key, key, 3) for conv1 / conv2 and (key, key, 1) for the downsample: a block
        # whose convolutions were configured differently (strided downsample, other kernel sizes, loaded variants) goes op by op
        want_ks = 3 if i < 2 else 1
        if (any(k != want_ks for k in c.kernel_size) or any(st != 1 for st in c.stride) or any(d != 1 for d in c.dilation)
                or c.kernel.dim() != (3 if want_ks == 3 else 2)):
            return False
    if convs[0].out_channels != convs[1].in_channels or (len(convs) == 3 and convs[2].out_channels != convs[1].out_channels):
        return False
    for n in norms:
        b = n.bn
        if (type(n) not in (_mod.MinkowskiBatchNorm, _mod.MinkowskiSyncBatchNorm) or type(n) is not type(norms[0])
                or not (b.training and b.affine and b.track_running_stats) or not _plain(n) or not _plain(b)):
            return False
    if type(norms[0]) is _mod.MinkowskiSyncBatchNorm and any(n.process_group is not norms[0].process_group for n in norms):
        return False
    return xf.dtype in (torch.bfloat16, torch.float32) and xf.shape[1] == blk.conv1.in_channels


def _sync_group(norm):
    """-> (True, process group) when this norm exchanges statistics across ranks in this call (MinkowskiSyncBatchNorm.forward's
    own condition), else (False, None)"""
    if type(norm) is not _mod.MinkowskiSyncBatchNorm:
        return False, None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False, None
    if dist.get_world_size(norm.process_group) > 1 or _mod.MinkowskiSyncBatchNorm.force_sync:
        return True, norm.process_group
    return False, None


def _bn_fwd(be, x, bn, g, b, residual, relu, conv_stats, sync=(False, None)):
    """-> y, stats, inv_n (inv_n: device scalar 1 / global rows of a SyncBN layer, else None)"""
    if conv_stats is not None and (conv_stats[1] is None):
        conv_stats = None                                  # pivot convention of MinkowskiBatchNorm.forward
    if sync[0]:
        from ..ddp import sync_bn_forward
        return sync_bn_forward(be, x, g, b, residual, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum,
                               relu, sync[1], conv_stats)
    return _bn_fwd_local(be, x, bn, g, b, residual, relu, conv_stats) + (None,)


def _bn_fwd_local(be, x, bn, g, b, residual, relu, conv_stats):
    if conv_stats is not None:
        return be.bn_forward(x, g, b, bn.eps, bn.momentum, bn.running_mean, bn.running_var, residual, relu, bn.num_batches_tracked,
                             conv_stats=conv_stats)
    return be.bn_forward(x, g, b, bn.eps, bn.momentum, bn.running_mean, bn.running_var, residual, relu, bn.num_batches_tracked)


def _bn_bwd(be, x, y, dy, g, b, gp, bp, stats, relu_mode, want_res, need, sync=(False, None), inv_n=None):
    """-> dx, dres, d gamma, d beta (the last two as bucket-slot views when the parameters gp / bp own slots)"""
    from .modules import grad_slot_view
    if sync[0]:
        from ..ddp import sync_bn_backward
        dx, dres, dg, db, slots = sync_bn_backward(be, x, y, dy.contiguous(), g, b, stats, inv_n, relu_mode, want_res, sync[1], gp, bp)
        if slots:
            return dx, dres, dg, db
        return dx, dres, (dg.to(g.dtype) if need else None), (db.to(g.dtype) if need else None)
    gv = grad_slot_view(gp) if gp is not None else None
    bv = grad_slot_view(bp) if bp is not None else None
    if gv is None or bv is None:
        gv = bv = None
    dx, dres, dg, db = be.bn_backward(x, y, dy, g, b, stats, relu_mode, want_res, gv, bv)
    if gv is not None:
        return dx, dres, gv, bv
    return dx, dres, (dg.to(g.dtype) if need else None), (db.to(g.dtype) if need else None)


_BLOCK_C = _tuning.host("BLOCK_C") != 0
_BLOCK_C_VETO = None    # measurement hook (bench.py's per-launch instrumentation): callable(rows, cin, planes) -> True = enqueue this
#                         block call by call (same launches, same results), so that its conv launches can be bracketed from Python


def _c_block_ok(x, kmap3, kmap1, cin, planes, be):
    """the one-call-per-direction entry points (csrc/lgs_block.hip) serve blocks whose input is a plain contiguous tensor and whose
    dgrad shape has the accumulating epilogue (weight gradients: on the compute stream for small batches, else on the side stream
    from inside the engine call)"""
    if not (_BLOCK_C and x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32)):
        return False
    if getattr(be, "conv_bn_stats", False) or not hasattr(be, "block_forward"):
        return False
    if _BLOCK_C_VETO is not None and _BLOCK_C_VETO(x.shape[0], cin, planes):
        return False
    key = ("cblk", cin, planes, x.dtype)
    ok = kmap3._wsb.get(key)
    if ok is None:
        from .. import engine
        dt = engine.LGS_BF16 if x.dtype == torch.bfloat16 else engine.LGS_F32
        ok = kmap3._wsb[key] = bool(engine.lib().lgs_conv_dgrad_can_accumulate(kmap3.h, 0, cin, planes, dt))
    return ok


class _BasicBlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, kmap3, kmap1, w1, g1, b1, w2, g2, b2, wd=None, gd=None, bd=None):
        be = get_backend()
        n1, n2 = blk.norm1.bn, blk.norm2.bn
        pc1, pc2 = blk.conv1._cache_for(x), blk.conv2._cache_for(x)
        # MinkowskiSyncBatchNorm blocks (N > 1): the same node, its norms exchange their statistics through the process group
        # (ddp.sync_bn_forward / sync_bn_backward, the functions the module itself runs) -- call by call, the engine's one-call
        # path has no collectives inside
        sync = ctx.sync = _sync_group(blk.norm1)
        ctx.c_path = (not sync[0]) and g1.dtype == torch.float32 and _c_block_ok(x, kmap3, kmap1, w1.shape[1], w1.shape[2], be)
        if ctx.c_path:
            # ONE engine call for the whole block (same launches, same order: bit-identical to the sequence below)
            ctx.has_ds = wd is not None
            nd = blk.downsample[1].bn if ctx.has_ds else None
            pcd = blk.downsample[0]._cache_for(x) if ctx.has_ds else None
            relu = bool(blk.final_relu)
            o1, st1, y1, o2, st2, y2, od, std = be.block_forward(x, kmap3, kmap1, (w1, w2, wd), (pc1, pc2, pcd), (n1, n2, nd),
                                                                 ((g1, b1), (g2, b2), (gd, bd)), relu)
            ctx.kmap3, ctx.kmap1, ctx.relu, ctx.pc = kmap3, kmap1, relu, (pc1, pc2, pcd)
            ctx.params = tuple(t if isinstance(t, nn.Parameter) else None for t in (w1, g1, b1, w2, g2, b2, wd, gd, bd))
            if ctx.has_ds:
                ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, o1, st1, y1, o2, st2, y2, wd, gd, bd, od, std)
            else:
                ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, o1, st1, y1, o2, st2, y2)
            return y2
        # the conv epilogue also emits the next norm's statistics (off by default)
        want = bool(getattr(be, "conv_bn_stats", False)) and be.want_conv_bn_stats(x.shape[0], w1.shape[-1], x.element_size())
        o1, s1 = kmap3.conv_forward(x, w1, None, False, bn_pivot=n1.running_mean, want_bn_stats=True, pack_cache=pc1) if want else \
            (kmap3.conv_forward(x, w1, None, False, pack_cache=pc1), None)
        y1, st1, inv1 = _bn_fwd(be, o1, n1, g1, b1, None, True, s1, sync)
        o2, s2 = kmap3.conv_forward(y1, w2, None, False, bn_pivot=n2.running_mean, want_bn_stats=True, pack_cache=pc2) if want else \
            (kmap3.conv_forward(y1, w2, None, False, pack_cache=pc2), None)
        ctx.has_ds = wd is not None
        if ctx.has_ds:
            nd = blk.downsample[1].bn
            pcd = blk.downsample[0]._cache_for(x)
            od, sd = kmap1.conv_forward(x, wd, None, False, bn_pivot=nd.running_mean, want_bn_stats=True, pack_cache=pcd) if want else \
                (kmap1.conv_forward(x, wd, None, False, pack_cache=pcd), None)
            res, std, invd = _bn_fwd(be, od, nd, gd, bd, None, False, sd, sync)
        else:
            res, invd = x, None
        relu = bool(blk.final_relu)
        y2, st2, inv2 = _bn_fwd(be, o2, n2, g2, b2, res, relu, s2, sync)
        ctx.kmap3, ctx.kmap1, ctx.relu, ctx.pc = kmap3, kmap1, relu, (pc1, pc2, pcd if ctx.has_ds else None)
        ctx.params = tuple(t if isinstance(t, nn.Parameter) else None for t in (w1, g1, b1, w2, g2, b2, wd, gd, bd))
        inv = ((inv1, inv2, invd) if ctx.has_ds else (inv1, inv2)) if sync[0] else ()
        if ctx.has_ds:
            ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, o1, st1, y1, o2, st2, y2, wd, gd, bd, od, std, *inv)
        else:
            ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, o1, st1, y1, o2, st2, y2, *inv)
        return y2

    @staticmethod
    def backward(ctx, dy):
        from .modules import conv_weight_grad
        be = get_backend()
        sv = ctx.saved_tensors
        x, w1, g1, b1, w2, g2, b2, o1, st1, y1, o2, st2, y2 = sv[:13]
        need = ctx.needs_input_grad
        kmap3, (pc1, pc2, pcd) = ctx.kmap3, ctx.pc
        pw1, pg1, pb1, pw2, pg2, pb2, pwd, pgd, pbd = ctx.params
        # inputs: 0 x | 1 blk 2 kmap3 3 kmap1 | 4 w1 5 g1 6 b1 | 7 w2 8 g2 9 b2 | 10 wd 11 gd 12 bd
        if ctx.c_path and all(need[i] for i in (4, 5, 6, 7, 8, 9)) and (not ctx.has_ds or all(need[i] for i in (10, 11, 12))):
            extra = sv[13:18] if ctx.has_ds else (None, None, None, None, None)
            return be.block_backward(dy, sv[:13], extra, kmap3, ctx.kmap1, ctx.pc, ctx.params, ctx.relu, bool(need[0]))
        sync = ctx.sync
        n_base = 18 if ctx.has_ds else 13
        inv1, inv2, invd = (tuple(sv[n_base:]) + (None,))[:3] if sync[0] else (None, None, None)
        # norm2 (+ residual) (+ ReLU): mask from the saved output when there is a ReLU (a residual was added)
        dx2, dres, dg2, db2 = _bn_bwd(be, o2, y2 if ctx.relu else None, dy, g2, b2, pg2, pb2, st2, 1 if ctx.relu else 0, True,
                                      need[8] or need[9], sync, inv2)
        gw2 = conv_weight_grad(kmap3, y1, dx2, False, pw2, w2.shape, w2.dtype) if need[7] else None
        dy1 = kmap3.conv_dgrad(dx2, w2, False, pack_cache=pc2)
        dx1, _, dg1, db1 = _bn_bwd(be, o1, None, dy1, g1, b1, pg1, pb1, st1, 2, False, need[5] or need[6], sync, inv1)
        gw1 = conv_weight_grad(kmap3, x, dx1, False, pw1, w1.shape, w1.dtype) if need[4] else None
        gin = None
        if ctx.has_ds:
            wd, gd, bd, od, std = sv[13:18]
            dxd, _, dgd, dbd = _bn_bwd(be, od, None, dres, gd, bd, pgd, pbd, std, 0, False, need[11] or need[12], sync, invd)
            gwd = conv_weight_grad(ctx.kmap1, x, dxd, False, pwd, wd.shape, wd.dtype) if need[10] else None
            if need[0]:
                gin = kmap3.conv_dgrad(dx1, w1, False, pack_cache=pc1,
                                       accumulate_into=ctx.kmap1.conv_dgrad(dxd, wd, False, pack_cache=pcd))
            return gin, None, None, None, gw1, dg1, db1, gw2, dg2, db2, gwd, dgd, dbd
        if need[0]:
            gin = kmap3.conv_dgrad(dx1, w1, False, pack_cache=pc1, accumulate_into=dres)
        return gin, None, None, None, gw1, dg1, db1, gw2, dg2, db2


class BlockView:
    """the modules of one matched block, under the attribute names _BasicBlockFunction reads"""
    __slots__ = ("conv1", "norm1", "conv2", "norm2", "downsample", "final_relu", "cat_up")

    def __init__(self, conv1, norm1, conv2, norm2, downsample, final_relu, cat_up):
        self.conv1, self.norm1, self.conv2, self.norm2 = conv1, norm1, conv2, norm2
        self.downsample, self.final_relu, self.cat_up = downsample, final_relu, cat_up


_DEAD_MID = "this tensor was an intermediate of a MinkowskiSyncBatchNorm residual block that ran as ONE fused node (me/block.py), so " \
            "its value was never formed, and recomputing it would be a collective only this rank enters; set LGS_BLOCK_FUSED=0 (or " \
            "LGS_DEFER=0) to run the block call by call and keep every intermediate"


def _is3(op):
    m = op.mod
    return (op.kind == _d.CONV and type(m) is _mod.MinkowskiConvolution and m.kernel_volume == 27 and m.stride[0] == 1
            and not op.aux[2])


WAIT = -1


def try_block(q, i, n, final=True):
    """q[i] is a recorded convolution at the head of the queue.  If q[i:] starts with a complete residual block whose intermediates
    have no other consumer, run it as one node and return the number of queue entries it covered; 0 = it is not such a block;
    WAIT (only when not `final`) = it may still become one: calls that complete it -- the second convolution, `out += residual`,
    the in-place ReLU, the consumer that closes the last norm -- have not been recorded yet."""
    if not _BLOCK_FUSED:
        return 0
    more = 0 if final else WAIT
    c1 = q[i]
    if not c1.bs or c1.dropped:
        return 0
    if i + 1 >= n:
        return more
    n1 = q[i + 1]
    if not (n1.kind == _d.BN and n1.inp is c1.out and n1.residual is None and not n1.cat_up and n1.cat_into is None and c1.uses == 1):
        return 0
    if n1.uses == 0:
        return 0 if n1.dropped else more              # the norm is still open: its ReLU / its consumer are not recorded yet
    if not n1.relu or n1.uses != 1 or i + 2 >= n:
        return 0
    c2 = q[i + 2]
    if not (c2.kind == _d.CONV and c2.bs and c2.inp is n1.out):
        return 0
    if i + 3 >= n:
        return more
    x = c1.inp
    nxt = q[i + 3]
    ds = None
    if nxt.kind == _d.BN:
        n2, took = nxt, 4
        if n2.inp is not c2.out:
            return 0
        if n2.residual is None:
            # norm2 before its `+=` (the downsample branch may be recorded in between): nothing else may have touched it
            return more if (not n2.relu and n2.uses == 0 and not n2.dropped) else 0
        if n2.residual is not x:
            return 0
    elif nxt.kind == _d.CONV:
        cd = nxt
        md = cd.mod
        if not (type(md) is _mod.MinkowskiConvolution and md.kernel_volume == 1 and md.use_mm and cd.inp is x):
            return 0
        if i + 4 >= n:
            return more
        nd = q[i + 4]
        if not (nd.kind == _d.BN and nd.inp is cd.out and not nd.relu and nd.residual is None and not nd.cat_up and nd.cat_into is None
                and cd.uses == 1):
            return 0
        if i + 5 >= n:
            return more
        n2, took = q[i + 5], 6
        if not (n2.kind == _d.BN and n2.residual is nd.out and nd.uses == 1 and n2.inp is c2.out):
            return 0
        ds = (cd.mod, nd.mod)
    else:
        return 0
    if not (c2.uses == 1 and not n2.cat_up and n2.cat_into is None):
        return 0
    if n2.uses == 0 and not final and not n2.dropped:
        return WAIT                                   # its (optional) final ReLU may still be recorded
    ops = q[i:i + took]
    if not all(o.grad for o in ops) or x._op is not None:
        return 0
    for o in ops:
        m = o.mod
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return 0
        if o.kind == _d.BN and (m.bn._forward_hooks or m.bn._forward_pre_hooks):
            return 0
    blk = BlockView(c1.mod, n1.mod, c2.mod, n2.mod, ds, bool(n2.relu), 0)
    xf = x._F
    if not _block_fast_path_ok(blk, xf):
        return 0
    kmap3 = c1.aux[1]
    if c2.aux[1] is not kmap3:
        return 0
    if ds is not None:
        y = _BasicBlockFunction.apply(xf, blk, kmap3, q[i + 3].aux[1], blk.conv1.kernel, blk.norm1.bn.weight, blk.norm1.bn.bias,
                                      blk.conv2.kernel, blk.norm2.bn.weight, blk.norm2.bn.bias,
                                      ds[0].kernel, ds[1].bn.weight, ds[1].bn.bias)
    else:
        y = _BasicBlockFunction.apply(xf, blk, kmap3, None, blk.conv1.kernel, blk.norm1.bn.weight, blk.norm1.bn.bias,
                                      blk.conv2.kernel, blk.norm2.bn.weight, blk.norm2.bn.bias)
    out = n2.out
    if out is not None:
        out._F, out._op = y, None
    # the intermediates were never formed.  A caller that kept one gets it recomputed on first read (deferred.LazyBlock); under
    # MinkowskiSyncBatchNorm the recomputation would be a collective that only the reading rank enters, so there it raises
    if _sync_group(blk.norm1)[0]:
        mark = _d._Dead(_DEAD_MID)
    else:
        import weakref
        steps = [(_d.CONV, c1.mod, c1.aux, -1, weakref.ref(c1.out) if c1.out is not None else _none),
                 (_d.BN, n1.mod, True, 0, weakref.ref(n1.out) if n1.out is not None else _none),
                 (_d.CONV, c2.mod, c2.aux, 1, weakref.ref(c2.out) if c2.out is not None else _none)]
        if ds is not None:
            cd, nd = q[i + 3], q[i + 4]
            steps += [(_d.CONV, cd.mod, cd.aux, -1, weakref.ref(cd.out) if cd.out is not None else _none),
                      (_d.BN, nd.mod, False, 3, weakref.ref(nd.out) if nd.out is not None else _none)]
        mark = _d.LazyBlock(x, steps)
    for o in ops:
        if o is not n2 and o.out is not None:
            o.out._op = mark
    return took


def _none():
    return None
