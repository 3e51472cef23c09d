"""Res16UNet family on the engine's MinkowskiEngine-compatible surface.

The reference's own model files (/root/reference/models/res16unet.py, resnet.py, clip_models.py,
modules/resnet_block.py) load unchanged through the `MinkowskiEngine` alias package; this module is the
build's self-contained counterpart (the reference .py files do not travel to the GPU box).  It
reproduces, and tests/test_models_manifest.py pins against fixtures captured from the reference:
  * the module tree / state-dict keys and shapes (`conv0p1s1.kernel`, `block2.0.downsample.1.bn.weight`, ...)
    so released checkpoints load (lib/utils.py:17-45),
  * the dataflow of Res16UNetBase.forward (res16unet.py:196-270) incl. cat order (upsampled, skip),
  * BasicBlock.forward (resnet_block.py:41-57) with the whole norm -> (+residual) -> ReLU tail fused
    into one engine call,
  * the reference's momentum quirk: stem/down/up/downsample norms use config.bn_momentum (0.02) while the
    norms inside blocks keep 0.1 (resnet.py:106-123 never forwards bn_momentum to block()).
"""
import torch
import torch.nn as nn

from . import me as ME


def _conv(cin, cout, ks, stride=1, bias=False, D=3):
    return ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, dilation=1, bias=bias, dimension=D)


def _conv_tr(cin, cout, ks, stride, D=3):
    return ME.MinkowskiConvolutionTranspose(cin, cout, kernel_size=ks, stride=stride, dilation=1, bias=False, dimension=D)


class BasicBlock(nn.Module):
    """conv3-norm-relu-conv3-norm-(+residual)-relu  (resnet_block.py:9-57); `final_relu=False` is the
    reference's NoReluBlock (resnet_block.py:133-161) used by the representation models."""
    expansion = 1

    def __init__(self, inplanes, planes, downsample=None, bn_momentum=0.1, D=3, final_relu=True):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, D=D)
        self.norm1 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = _conv(planes, planes, 3, D=D)
        self.norm2 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.downsample = downsample
        self.inplanes, self.planes, self.final_relu = inplanes, planes, final_relu
        self.cat_up = 0          # > 0: this block's output is a skip tensor; channels of the `up` half it will be concatenated with

    def forward(self, x):
        if _block_fast_path_ok(self, x):
            return _block_fast_forward(self, x)
        out = self.norm1(self.conv1(x, bn=self.norm1), relu=True)
        out = self.conv2(out, bn=self.norm2)
        residual = x if self.downsample is None else self.downsample[1](self.downsample[0](x, bn=self.downsample[1]))
        return self.norm2(out, relu=self.final_relu, residual=residual, cat_up=self.cat_up)


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
from . import tuning as _tuning
_BLOCK_FUSED = _tuning.host("BLOCK_FUSED") != 0


def _plain(m):
    return not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


def _block_fast_path_ok(blk, x):
    if not _BLOCK_FUSED or blk.cat_up or not x.F.is_cuda or not torch.is_grad_enabled():
        return False
    from torch.nn.modules import module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks or not _plain(blk):
        return False
    be = ME.get_backend()
    if not (hasattr(be, "bn_forward") and hasattr(be, "side_stream") and getattr(be, "bn_counts_batches", False)):
        return False
    convs, norms = [blk.conv1, blk.conv2], [blk.norm1, blk.norm2]
    if blk.downsample is not None:
        if len(blk.downsample) != 2:
            return False
        convs.append(blk.downsample[0]); norms.append(blk.downsample[1])
    for i, c in enumerate(convs):
        if type(c) is not ME.MinkowskiConvolution or c.bias is not None or c.kernel.dtype != torch.float32 or not _plain(c):
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
        if (type(n) not in (ME.MinkowskiBatchNorm, ME.MinkowskiSyncBatchNorm) or type(n) is not type(norms[0])
                or not (b.training and b.affine and b.track_running_stats) or not _plain(n) or not _plain(b)):
            return False
    if type(norms[0]) is ME.MinkowskiSyncBatchNorm and any(n.process_group is not norms[0].process_group for n in norms):
        return False
    return x.F.dtype in (torch.bfloat16, torch.float32) and x.F.shape[1] == blk.conv1.in_channels


def _sync_group(norm):
    """-> (True, process group) when this norm exchanges statistics across ranks in this call (MinkowskiSyncBatchNorm.forward's
    own condition), else (False, None)"""
    if type(norm) is not ME.MinkowskiSyncBatchNorm:
        return False, None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False, None
    if dist.get_world_size(norm.process_group) > 1 or ME.MinkowskiSyncBatchNorm.force_sync:
        return True, norm.process_group
    return False, None


def _block_fast_forward(blk, x):
    mgr, key = x.coordinate_manager, x.coordinate_map_key
    kmap3 = mgr.kernel_map_handle(key, key, 3)
    ds = blk.downsample
    if ds is not None:
        kmap1 = mgr.kernel_map_handle(key, key, 1)
        y = _BasicBlockFunction.apply(x.F, blk, kmap3, kmap1, blk.conv1.kernel, blk.norm1.bn.weight, blk.norm1.bn.bias,
                                      blk.conv2.kernel, blk.norm2.bn.weight, blk.norm2.bn.bias,
                                      ds[0].kernel, ds[1].bn.weight, ds[1].bn.bias)
    else:
        y = _BasicBlockFunction.apply(x.F, blk, kmap3, None, blk.conv1.kernel, blk.norm1.bn.weight, blk.norm1.bn.bias,
                                      blk.conv2.kernel, blk.norm2.bn.weight, blk.norm2.bn.bias)
    return ME.SparseTensor(y, coordinate_map_key=key, coordinate_manager=mgr)


def _bn_fwd(be, x, bn, g, b, residual, relu, conv_stats, sync=(False, None)):
    """-> y, stats, inv_n (inv_n: device scalar 1 / global rows of a SyncBN layer, else None)"""
    if conv_stats is not None and (conv_stats[1] is None):
        conv_stats = None                                  # pivot convention of MinkowskiBatchNorm.forward
    if sync[0]:
        from .ddp import sync_bn_forward
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
    from .me.modules import grad_slot_view
    if sync[0]:
        from .ddp import sync_bn_backward
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
        from . import engine
        dt = engine.LGS_BF16 if x.dtype == torch.bfloat16 else engine.LGS_F32
        ok = kmap3._wsb[key] = bool(engine.lib().lgs_conv_dgrad_can_accumulate(kmap3.h, 0, cin, planes, dt))
    return ok


class _BasicBlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, kmap3, kmap1, w1, g1, b1, w2, g2, b2, wd=None, gd=None, bd=None):
        be = ME.get_backend()
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
        from .me.modules import conv_weight_grad
        be = ME.get_backend()
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


class Res16UNet(ME.MinkowskiNetwork):
    """4x (stride-2 conv + residual stage) down, 4x (transposed conv + skip concat + residual stage) up,
    1x1 classifier.  PLANES/LAYERS per variant below."""
    PLANES = (32, 64, 128, 256, 256, 256, 256, 256)
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
    INIT_DIM = 32
    NO_FINAL_RELU = False   # representation variants: last block of block8 has no ReLU (clip_models.py:101)

    def __init__(self, in_channels, out_channels, config=None, D=3, **kwargs):
        super().__init__(D)
        self.in_channels, self.out_channels, self.config = in_channels, out_channels, config
        bn_m = getattr(config, "bn_momentum", 0.02) if config is not None else 0.02
        k0 = getattr(config, "conv1_kernel_size", 3) if config is not None else 3
        P, Lr = self.PLANES, self.LAYERS
        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = _conv(in_channels, self.inplanes, k0, D=D)
        self.bn0 = ME.MinkowskiBatchNorm(self.inplanes, momentum=bn_m)
        names_down = ["conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"]
        for i in range(4):
            setattr(self, names_down[i], _conv(self.inplanes, self.inplanes, 2, stride=2, D=D))
            setattr(self, "bn%d" % (i + 1), ME.MinkowskiBatchNorm(self.inplanes, momentum=bn_m))
            setattr(self, "block%d" % (i + 1), self._make_layer(P[i], Lr[i], bn_m))
        names_up = ["convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"]
        skips = [P[2], P[1], P[0], self.INIT_DIM]
        for j in range(4):
            i = 4 + j
            setattr(self, names_up[j], _conv_tr(self.inplanes, P[i], 2, 2, D=D))
            setattr(self, "bntr%d" % i, ME.MinkowskiBatchNorm(P[i], momentum=bn_m))
            self.inplanes = P[i] + skips[j] * BasicBlock.expansion
            last = (j == 3)
            setattr(self, "block%d" % (i + 1), self._make_layer(P[i], Lr[i], bn_m, no_final_relu=last and self.NO_FINAL_RELU))
        self.final = _conv(P[7], out_channels, 1, bias=True, D=D)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.repr_only = False
        # zero-copy ME.cat: the last block of block1..3 (and bn0) produce the skip tensors of convtr6 / 5 / 4 (and 7)
        self.block1[-1].cat_up, self.block2[-1].cat_up, self.block3[-1].cat_up = P[6], P[5], P[4]
        self._cat_up0 = P[7]

    def _make_layer(self, planes, blocks, bn_momentum, no_final_relu=False):
        downsample = None
        if self.inplanes != planes * BasicBlock.expansion:
            downsample = nn.Sequential(_conv(self.inplanes, planes * BasicBlock.expansion, 1, D=self.D),
                                       ME.MinkowskiBatchNorm(planes * BasicBlock.expansion, momentum=bn_momentum))
        layers = [BasicBlock(self.inplanes, planes, downsample=downsample, D=self.D,
                             final_relu=not (no_final_relu and blocks == 1))]
        self.inplanes = planes * BasicBlock.expansion
        for b in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes, D=self.D, final_relu=not (no_final_relu and b == blocks - 1)))
        return nn.Sequential(*layers)

    def representation_only(self, flag):
        """clip_models.py:106-109: drop the classifier, return the block8 features only."""
        self.repr_only = flag
        self.final = None

    def _prefetch_maps(self, x):
        """Ask for every coordinate / kernel map of the U-Net up front: they are built on the engine's map stream, so the four
        coarsenings and nine kernel maps overlap the first convolutions instead of each being requested just in time by the first
        layer of its level (one 145 k-voxel scene per step: 12.0 -> 10.5 ms; no effect on the large batch)."""
        mgr, key = x.coordinate_manager, x.coordinate_map_key
        for lvl in range(5):
            mgr.kernel_map_handle(key, key, 3)
            if lvl < 4:
                nk = mgr.stride(key, 2)
                mgr.kernel_map_handle(key, nk, 2)
                key = nk

    def trunk(self, x):
        if x.F.is_cuda:
            self._prefetch_maps(x)
        # conv(x, bn=norm): the conv epilogue hands the norm its batch statistics (me.modules.MinkowskiConvolutionBase.forward)
        out_p1 = self.bn0(self.conv0p1s1(x, bn=self.bn0), relu=True, cat_up=self._cat_up0)
        out_b1p2 = self.block1(self.bn1(self.conv1p1s2(out_p1, bn=self.bn1), relu=True))
        out_b2p4 = self.block2(self.bn2(self.conv2p2s2(out_b1p2, bn=self.bn2), relu=True))
        out_b3p8 = self.block3(self.bn3(self.conv3p4s2(out_b2p4, bn=self.bn3), relu=True))
        out = self.block4(self.bn4(self.conv4p8s2(out_b3p8, bn=self.bn4), relu=True))
        out = self.block5(ME.cat(self.bntr4(self.convtr4p16s2(out, bn=self.bntr4), relu=True, cat_into=out_b3p8), out_b3p8))
        out = self.block6(ME.cat(self.bntr5(self.convtr5p8s2(out, bn=self.bntr5), relu=True, cat_into=out_b2p4), out_b2p4))
        out = self.block7(ME.cat(self.bntr6(self.convtr6p4s2(out, bn=self.bntr6), relu=True, cat_into=out_b1p2), out_b1p2))
        out = self.block8(ME.cat(self.bntr7(self.convtr7p2s2(out, bn=self.bntr7), relu=True, cat_into=out_p1), out_p1))
        return out

    def forward(self, x):
        out = self.trunk(x)
        if self.repr_only:
            return out
        return self.final(out), out


class Res16UNet14(Res16UNet):
    LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class Res16UNet18(Res16UNet):
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)


class Res16UNet34(Res16UNet):
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet14A(Res16UNet14):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class Res16UNet18A(Res16UNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class Res16UNet34A(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 64)


class Res16UNet34B(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 32)


class Res16UNet34C(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)


class Res16UNet34CR(Res16UNet34C):
    """34C with a ReLU-free last block and representation_only() (clip_models.py:94-183)."""
    NO_FINAL_RELU = True


class Res16UNet34CR_Proj(Res16UNet34CR):
    """+ learned projection of the 512-d CLIP anchors to PLANES[7] (clip_models.py:186-200)."""

    def __init__(self, in_channels, out_channels, config=None, D=3, **kwargs):
        super().__init__(in_channels, out_channels, config, D)
        self.projection_layer = nn.Conv1d(512, self.PLANES[7], kernel_size=1, stride=1, bias=True)

    def forward(self, x, anchor_feats):
        return super().forward(x), self.projection_layer(anchor_feats.unsqueeze(-1)).squeeze()


class Res16UNet34D(Res16UNet34CR):
    """CLIP-dimensional representation model, default of text_representation_train.sh:7 (clip_models.py:205-215)."""
    PLANES = (32, 64, 128, 256, 256, 256, 256, 512)


class _InsSegHead:
    """downstream/insseg head on the same trunk (insseg_models/insseg_res16unet.py:197-199,260-265):
    offsets = 1x1(C->3, bias)(ReLU(BN(1x1(C->C, bias)(block8 features)))); forward -> (offsets, logits, features).

    freeze_trunk(True) is BASELINE configs[4]'s framing ("instance-seg head on FROZEN pretrained features"; the reference
    itself optimises all parameters, downstream/insseg/lib/pl_Trainer.py:81): the U-Net trunk stops requiring gradients,
    its norms use their running statistics (eval mode, whatever .train() is called on the model), and the trunk runs
    under no_grad -- the step is then trunk forward + head forward/backward (offsets_pre, bntr_offset, offsets, final)."""

    HEAD = ("offsets_pre", "bntr_offset", "offsets", "final")

    def _add_head(self, D):
        c = self.PLANES[7]
        bn_m = self.bn0.bn.momentum
        self.offsets_pre = _conv(c, c, 1, bias=True, D=D)
        self.bntr_offset = ME.MinkowskiBatchNorm(c, momentum=bn_m)
        self.offsets = _conv(c, 3, 1, bias=True, D=D)
        self.trunk_frozen = False

    def _trunk_modules(self):
        return [m for n, m in self.named_children() if n not in self.HEAD]

    def freeze_trunk(self, flag=True):
        self.trunk_frozen = bool(flag)
        for m in self._trunk_modules():
            for p in m.parameters():
                p.requires_grad_(not flag)
        self.train(self.training)
        return self

    def train(self, mode=True):
        super().train(mode)
        if getattr(self, "trunk_frozen", False):
            for m in self._trunk_modules():
                m.eval()                                  # frozen features: running statistics, no updates
        return self

    def forward(self, x, detach=False):
        if getattr(self, "trunk_frozen", False):
            with torch.no_grad():
                out = self.trunk(x)
        else:
            out = self.trunk(x)
        off = self.offsets(self.bntr_offset(self.offsets_pre(out), relu=True))
        return off, self.final(out), out


class InsSegRes16UNet14A(_InsSegHead, Res16UNet14A):
    def __init__(self, in_channels, out_channels, config=None, D=3, **kwargs):
        Res16UNet14A.__init__(self, in_channels, out_channels, config, D, **kwargs)
        self._add_head(D)


class InsSegRes16UNet34C(_InsSegHead, Res16UNet34C):
    def __init__(self, in_channels, out_channels, config=None, D=3, **kwargs):
        Res16UNet34C.__init__(self, in_channels, out_channels, config, D, **kwargs)
        self._add_head(D)


MODELS = {c.__name__: c for c in [InsSegRes16UNet14A, InsSegRes16UNet34C,
                                  Res16UNet14, Res16UNet18, Res16UNet34, Res16UNet14A, Res16UNet18A, Res16UNet34A,
                                  Res16UNet34B, Res16UNet34C, Res16UNet34CR, Res16UNet34CR_Proj, Res16UNet34D]}


def load_model(name):
    """same contract as /root/reference/models/__init__.py load_model(name) -> class"""
    if name not in MODELS:
        raise ValueError("unknown model %s (available: %s)" % (name, sorted(MODELS)))
    return MODELS[name]


def cast_features(sinput, dtype):
    """bf16 storage for features (weights stay fp32 masters; accumulation and BN statistics are fp32)."""
    return sinput if sinput.F.dtype == dtype else sinput._like(sinput.F.to(dtype))
