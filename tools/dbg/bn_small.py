"""stand-alone time of one BatchNorm direction on coarse-level tensors: the one-launch (grid barrier) kernels vs the three-launch
path.  GPU box: PYTHONPATH=$GRAFT_REPO_ROOT python tools/dbg/bn_small.py"""
import torch
import MinkowskiEngine as ME
from languagegroundedsemseg_amd import engine

DEV = "cuda:0"


def timeit(fn, it=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    be = ME.get_backend()
    for n, c in ((327170, 32), (81023, 64), (81023, 128), (19643, 128), (19643, 256), (4985, 256), (2500, 256), (600, 256)):
        f = torch.randn(n, c, device=DEV).to(torch.bfloat16)
        g1, b1 = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        dy = torch.randn_like(f)
        out = []
        for fold, fused in ((1, 1), (0, 1), (0, 0)):
            with engine.tuning(BN_FOLD=fold, BN_FUSED=fused, BN_FUSED_FWD_MAX_MB=24 if fused else 0):
                y, st = be.bn_forward(f, g1, b1, 1e-5, 0.1, rm, rv, None, 1)
                tf = timeit(lambda: be.bn_forward(f, g1, b1, 1e-5, 0.1, rm, rv, None, 1))
                tb2 = timeit(lambda: be.bn_backward(f, y, dy, g1, b1, st, 2, False))
                tb1 = timeit(lambda: be.bn_backward(f, y, dy, g1, b1, st, 1, True))
                out.append((tf, tb2, tb1))
        print("rows %6d x %3d ch (%5.1f MB): two launches (fold in apply) fwd %5.1f us  bwd(relu from x) %5.1f  bwd(res) %5.1f | one launch "
              "(grid barriers) %5.1f  %5.1f  %5.1f | three launches %5.1f  %5.1f  %5.1f" % (n, c, n * c * 2 / 1e6, *out[0], *out[1], *out[2]))


main()
