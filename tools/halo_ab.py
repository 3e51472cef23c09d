"""A/B of k_conv_halo vs k_conv_gather on the benchmark's level-0 / level-1 shapes (stand-alone launches, HIP events).
usage: python tools/halo_ab.py [scenes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import MinkowskiEngine as ME
from languagegroundedsemseg_amd import engine
from languagegroundedsemseg_amd.synthetic import make_batch
from microbench import timeit

DEV = "cuda:0"


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    coords, feats, _ = make_batch(list(range(B)), n_target=150000, shift_seed=0)
    c = torch.from_numpy(coords).to(DEV)
    n = coords.shape[0]

    def maps(halo):
        with engine.tuning(HALO=halo):
            x = ME.SparseTensor(torch.zeros(n, 3, device=DEV).bfloat16(), c)
            m, k = x.coordinate_manager, x.coordinate_map_key
            out = [(m, k, m.kernel_map_handle(k, k, 3))]
            k1 = m.stride(k, 2)
            out.append((m, k1, m.kernel_map_handle(k1, k1, 3)))
            k2 = m.stride(k1, 2)
            out.append((m, k2, m.kernel_map_handle(k2, k2, 3)))
        return out
    for halo in (0, 1):
        t = timeit(lambda: maps(halo), 3, 1)
        print("maps L0..L2 (insert + 2 strides + 3x 3^3) halo=%d: %.2f ms" % (halo, t))
    mh, mg = maps(1), maps(0)
    torch.cuda.synchronize()
    for lvl, shapes in ((0, ((96, 96), (128, 96), (96, 128))), (1, ((32, 32), (96, 96), (128, 96))), (2, ((64, 64), (128, 128), (192, 128)))):
        rows = mh[lvl][0].size(mh[lvl][1])
        kk, _, _ = mg[lvl][2].export()
        M = kk.shape[0]
        for cin, cout in shapes:
            f = torch.randn(rows, cin, device=DEV).bfloat16()
            g = torch.randn(rows, cout, device=DEV).bfloat16()
            w = torch.randn(27, cin, cout, device=DEV) * 0.05
            line = "L%d rows %7d pairs %8d  %3d->%3d " % (lvl, rows, M, cin, cout)
            outs = []
            for name, ms in (("halo", mh), ("gather", mg)):
                km = ms[lvl][2]
                with engine.tuning(HALO=1 if name == "halo" else 0):
                    engine.dispatch_counts(reset=True)
                    o = km.conv_forward(f, w, None, False)
                    used = [k.split("<")[0] for k in engine.dispatch_counts() if k.startswith("k_conv")]
                    tf = timeit(lambda: km.conv_forward(f, w, None, False), 10, 3)
                    td = timeit(lambda: km.conv_dgrad(g, w, False), 10, 3)
                outs.append(o.float())
                balg = M * cin * 2 + rows * cout * 2 + 8 * M + 27 * cin * cout * 2
                line += " | %s(%s) fwd %.3f ms (%.2f TB/s alg) dgrad %.3f" % (name, ",".join(sorted(set(used))), tf, balg / tf / 1e9, td)
            d = float((outs[0] - outs[1]).norm() / outs[1].norm())
            print(line + " | rel diff %.2e" % d, flush=True)
            if os.environ.get("HALO_AB_TRACE"):
                with engine.tuning(HALO=1, HALO_TRACE=1):
                    mh[lvl][2].conv_forward(f, w, None, False)
                    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
