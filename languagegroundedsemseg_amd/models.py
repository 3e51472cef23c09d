"""Res16UNet family on the engine's MinkowskiEngine-compatible surface.

The reference's own model files (/root/reference/models/res16unet.py, resnet.py, clip_models.py,
modules/resnet_block.py) load unchanged through the `MinkowskiEngine` alias package; this module is the
build's self-contained counterpart (the reference .py files do not travel to the GPU box).  It
reproduces, and tests/test_models_manifest.py pins against fixtures captured from the reference:
  * the module tree / state-dict keys and shapes (`conv0p1s1.kernel`, `block2.0.downsample.1.bn.weight`, ...)
    so that a released checkpoint's tensors find their slots by name and shape (lib/utils.py:17-45).  UNTESTED, and not a
    claim that such a checkpoint reproduces its accuracy here: the order of the K kernel offsets inside `*.kernel [K, Cin, Cout]`
    (first spatial axis fastest) is recalled from MinkowskiEngine 0.5.4, whose source is not in /root/reference and
    could not be checked in this build -- no ME, no checkpoint, no network,
  * the dataflow of Res16UNetBase.forward (res16unet.py:196-270) incl. cat order (upsampled, skip),
  * BasicBlock.forward (resnet_block.py:41-57) CALL FOR CALL -- norm(x); relu(x) in place; out += residual; relu; me.cat --
    with standard MinkowskiEngine signatures only: the fusion (norm + residual + ReLU in one kernel, zero-copy cat, one engine
    call per block and direction) happens behind the ME surface (me/deferred.py), so the reference's unchanged files get
    exactly the launches this module gets,
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

    def forward(self, x):
        # call for call what resnet_block.py:41-57 (and NoReluBlock :147-161) issues; the ME surface records these calls and runs
        # them fused (me/deferred.py): norm + ReLU, norm + residual + ReLU, or the whole block as one engine call per direction
        residual = x
        out = self.conv1(x)
        out = self.norm1(out)
        out = self.relu(out)
        out = self.conv2(out)
        out = self.norm2(out)
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        if self.final_relu:
            out = self.relu(out)
        return out


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
        """res16unet.py:196-270, call for call (standard MinkowskiEngine signatures only)"""
        if x.F.is_cuda:
            self._prefetch_maps(x)
        out = self.conv0p1s1(x)
        out = self.bn0(out)
        out_p1 = self.relu(out)

        out = self.conv1p1s2(out_p1)
        out = self.bn1(out)
        out = self.relu(out)
        out_b1p2 = self.block1(out)

        out = self.conv2p2s2(out_b1p2)
        out = self.bn2(out)
        out = self.relu(out)
        out_b2p4 = self.block2(out)

        out = self.conv3p4s2(out_b2p4)
        out = self.bn3(out)
        out = self.relu(out)
        out_b3p8 = self.block3(out)

        out = self.conv4p8s2(out_b3p8)
        out = self.bn4(out)
        out = self.relu(out)
        out = self.block4(out)

        out = self.convtr4p16s2(out)
        out = self.bntr4(out)
        out = self.relu(out)
        out = ME.cat(out, out_b3p8)
        out = self.block5(out)

        out = self.convtr5p8s2(out)
        out = self.bntr5(out)
        out = self.relu(out)
        out = ME.cat(out, out_b2p4)
        out = self.block6(out)

        out = self.convtr6p4s2(out)
        out = self.bntr6(out)
        out = self.relu(out)
        out = ME.cat(out, out_b1p2)
        out = self.block7(out)

        out = self.convtr7p2s2(out)
        out = self.bntr7(out)
        out = self.relu(out)
        out = ME.cat(out, out_p1)
        out = self.block8(out)
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
        offsets = self.offsets_pre(out)                   # insseg_res16unet.py:260-265
        offsets = self.bntr_offset(offsets)
        offsets = self.relu(offsets)
        offsets = self.offsets(offsets)
        return offsets, self.final(out), out


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
