"""Per-op timings on the GPU (HIP events on torch's current stream) for the hot layer shapes.
usage: python tools/microbench.py [scenes]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import MinkowskiEngine as ME
from languagegroundedsemseg_amd.synthetic import make_batch

DEV = "cuda:0"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    coords, feats, labels = make_batch(list(range(B)), n_target=150000, shift_seed=0)
    if os.environ.get("LGS_MB_SORT") == "1":   # experiment: spatially sorted input rows instead of the dataset's arbitrary order
        from languagegroundedsemseg_amd.synthetic import morton_order
        perm = morton_order(coords)
        coords, feats, labels = coords[perm], feats[perm], labels[perm]
        print("input rows in Morton order")
    c = torch.from_numpy(coords).to(DEV)
    n = coords.shape[0]
    print("voxels", n)
    t0 = time.time()
    x = ME.SparseTensor(torch.from_numpy(feats).to(DEV), c)
    torch.cuda.synchronize()
    print("insert (first call, incl. lib load) %.1f ms" % ((time.time() - t0) * 1e3))

    def build_maps():
        xx = ME.SparseTensor(torch.zeros(n, 3, device=DEV), c)
        m = xx.coordinate_manager
        k = xx.coordinate_map_key
        for lvl in range(5):
            m.kernel_map_handle(k, k, 3)
            if lvl < 4:
                k2 = m.stride(k, 2)
                m.kernel_map_handle(k, k2, 2)
                k = k2
    print("all maps (5x 3^3 + 4x 2^3 + 4 strides + insert): %.2f ms" % timeit(build_maps, 3, 1))

    mgr, k0 = x.coordinate_manager, x.coordinate_map_key
    km = mgr.kernel_map_handle(k0, k0, 3)
    kk, ii, oo = km.export()
    M = kk.shape[0]
    print("3^3 pairs at L0: %d (%.2f per voxel)" % (M, M / n))
    for dtype in (torch.bfloat16, torch.float32):
        e = 2 if dtype == torch.bfloat16 else 4
        for cin, cout in ((96, 96), (128, 96), (32, 32)):
            f = torch.randn(n, cin, device=DEV).to(dtype)
            g = torch.randn(n, cout, device=DEV).to(dtype)
            w = torch.randn(27, cin, cout, device=DEV) * 0.05
            tf = timeit(lambda: km.conv_forward(f, w, None, False))
            td = timeit(lambda: km.conv_dgrad(g, w, False))
            tw = timeit(lambda: km.conv_wgrad(f, g, False))
            flop = 2.0 * M * cin * cout
            bf = M * cin * e + n * cout * e + 8 * M + 27 * cin * cout * e
            print("%-9s %3d->%3d  fwd %.3f ms (%.1f TF, %.2f TB/s alg)  dgrad %.3f ms  wgrad %.3f ms (%.1f TF)" % (
                str(dtype).split(".")[1], cin, cout, tf, flop / tf / 1e9, bf / tf / 1e9, td, tw, flop / tw / 1e9))
        f = torch.randn(n, 96, device=DEV).to(dtype)
        bn = ME.MinkowskiBatchNorm(96).to(DEV)
        sx = ME.SparseTensor(f.clone().requires_grad_(True), coordinate_map_key=k0, coordinate_manager=mgr)
        tb = timeit(lambda: bn(sx, relu=True))
        print("%-9s BN+ReLU fwd 96ch %.3f ms (%.2f TB/s of 3*N*C*e)" % (str(dtype).split(".")[1], tb, 3 * n * 96 * e / tb / 1e9))
        be = ME.get_backend()
        g1, b1 = torch.ones(96, device=DEV), torch.zeros(96, device=DEV)
        rm, rv = torch.zeros(96, device=DEV), torch.ones(96, device=DEV)
        y, st = be.bn_forward(f, g1, b1, 1e-5, 0.1, rm, rv, None, 1)
        dy = torch.randn_like(f)
        for mode, want in ((2, False), (1, True)):
            t_all = timeit(lambda: be.bn_backward(f, y, dy, g1, b1, st, mode, want))
            t_red = timeit(lambda: be.bn_backward_reduce(f, y, dy, g1, b1, st, mode))
            sums = be.bn_backward_reduce(f, y, dy, g1, b1, st, mode)
            t_app = timeit(lambda: be.bn_backward_apply(f, y, dy, g1, b1, st, sums, 1.0 / n, mode, want))
            nt = (5 if mode == 2 else 7) + (1 if want else 0)
            print("%-9s BN bwd 96ch relu-mode %d res %d: %.3f ms (%.2f TB/s of %d*N*C*e)  reduce %.3f  apply %.3f" % (
                str(dtype).split(".")[1], mode, want, t_all, nt * n * 96 * e / t_all / 1e9, nt, t_red, t_app))


def clip():
    """MFMA sub-report: S = normalize(F) . normalize(T)^T, [N,C] x [C,200]"""
    n = 1200000
    for c in (512, 96):
        for dtype in (torch.bfloat16, torch.float32):
            f = torch.randn(n, c, device=DEV).to(dtype)
            t = torch.randn(200, c, device=DEV)
            be = ME.get_backend()
            ms = timeit(lambda: be.clip_similarity(f, t))
            flop = 2.0 * n * c * 200
            byts = n * c * f.element_size() + n * 200 * 4
            print("clip similarity N=%d C=%d %-8s %.3f ms  %.1f TFLOP/s  %.2f TB/s (read F + write S)" % (
                n, c, str(dtype).split(".")[1], ms, flop / ms / 1e9, byts / ms / 1e9))
            lab = torch.randint(-1, 200, (n,), device=DEV)
            neg = torch.randint(0, 200, (n, 3), device=DEV)
            ms = timeit(lambda: be.clip_loss_forward(f, t, lab, neg, -1))
            byts = n * c * f.element_size() + n * (3 * 4 + 8 + 4 * 8)
            print("fused clip loss fwd (d_pos, d_neg, argmax, 1/|f|; no S) %.3f ms  %.1f TFLOP/s (%.1f %% of 2.5 PF)  %.2f TB/s "
                  "(%.1f %% of 8 TB/s)" % (ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0 * 100.0, byts / ms / 1e9, byts / ms / 1e9 / 8.0 * 100.0))
            d_pos, d_neg, pred, saved, _ = be.clip_loss_forward(f, t, lab, neg, -1)
            gp, gn = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
            ms = timeit(lambda: be.clip_loss_backward(saved, d_pos, d_neg, gp, gn, -1))
            byts = 2 * n * c * f.element_size()
            print("fused clip loss bwd (4-sparse upstream) %.3f ms  %.2f TB/s (read F + write gF)" % (ms, byts / ms / 1e9))


def quantize():
    """raw points -> voxel coordinates on the device (voxelize + dedup + label vote)"""
    import numpy as np
    coords, feats, labels = make_batch(list(range(8)), n_target=150000, shift_seed=0)
    rng = np.random.default_rng(0)
    rep = np.repeat(coords, 2, 0)                                   # ~2 points per voxel, like a 2 cm voxelisation of ScanNet
    pts = (rep[:, 1:].astype(np.float64) + rng.uniform(0.05, 0.95, (rep.shape[0], 3))) * 0.02
    p = torch.from_numpy(pts.astype(np.float32)).to(DEV)
    lab = torch.from_numpy(np.repeat(labels, 2)).to(DEV)
    n = p.shape[0]
    tv = timeit(lambda: ME.utils.voxelize(p, quantization_size=0.02))
    tq = timeit(lambda: ME.utils.sparse_quantize(p, None, lab, ignore_label=-1, quantization_size=0.02, return_index=True))
    print("voxelize %d points: %.3f ms (%.2f TB/s of 28 B/point)" % (n, tv, n * 28 / tv / 1e9))
    print("sparse_quantize (voxelize + dedup + label vote) %d points -> voxels: %.3f ms (%.1f M points/s)" % (n, tq, n / tq / 1e3))
    import time
    m = 300000                                                      # bounded CPU sample of the same workload (host path = numpy)
    ph, lh = p[:m].cpu(), lab[:m].cpu()
    t0 = time.perf_counter()
    ME.utils.sparse_quantize(ph, None, lh, ignore_label=-1, quantization_size=0.02, return_index=True)
    tc = time.perf_counter() - t0
    print("host path (numpy, 1 core) on the first %d points: %.1f ms (%.2f M points/s)" % (m, tc * 1e3, m / tc / 1e6))


def insseg():
    """instance-seg model (trunk + offset head), CE + offset losses, forward + backward, bf16"""
    import numpy as np
    from languagegroundedsemseg_amd import models
    from languagegroundedsemseg_amd.losses import fused_cross_entropy, instance_offset_losses
    coords, feats, labels = make_batch(list(range(8)), n_target=150000, shift_seed=0)
    c = torch.from_numpy(coords).to(DEV)
    f = torch.from_numpy(feats).to(DEV).bfloat16()
    lab = torch.from_numpy(labels).to(DEV)
    rng = np.random.default_rng(0)
    inst = torch.from_numpy(rng.integers(-1, 30, coords.shape[0])).to(DEV)
    centers = (c[:, 1:].float() + torch.randn(coords.shape[0], 3, device=DEV) * 20)

    class Cfg:
        bn_momentum, conv1_kernel_size = 0.02, 3
    m = models.load_model("InsSegRes16UNet34C")(3, 200, Cfg()).to(DEV).train()

    def step():
        for p in m.parameters():
            p.grad = None
        x = ME.SparseTensor(f, c)
        off, logits, _ = m(x)
        nl, dl = instance_offset_losses(off.F, c[:, 1:], centers, inst, 0.02)
        (fused_cross_entropy(logits.F, lab, ignore_index=-1) + nl + dl).backward()
    t = timeit(step)
    print("InsSegRes16UNet34C fwd+bwd (CE + offset losses, incl. map build) %d voxels: %.2f ms = %.1f M voxels/s" % (
        coords.shape[0], t, coords.shape[0] / t / 1e3))


def cluster():
    """PointGroup proposal clustering of one scene (ball query radius 3 cm + same-label components)"""
    import time
    import numpy as np
    from languagegroundedsemseg_amd.pointgroup import cluster_points
    from oracle import oracle as orc
    coords, feats, labels = make_batch([0], n_target=150000, shift_seed=0)
    rng = np.random.default_rng(0)
    xyz = (coords[:, 1:].astype(np.float32) + rng.uniform(0.2, 0.8, (coords.shape[0], 3)).astype(np.float32)) * np.float32(0.02)
    blk = np.floor(xyz / np.float32(0.4)).astype(np.int64)
    sem = ((blk[:, 0] * 3 + blk[:, 1] * 5 + blk[:, 2] * 7) % 7).astype(np.int32)
    x, s = torch.from_numpy(xyz).to(DEV), torch.from_numpy(sem).to(DEV)
    t = timeit(lambda: cluster_points(x, s, 0.03, 50))
    idx, off = cluster_points(x, s, 0.03, 50)
    print("cluster_points %d points -> %d clusters: %.3f ms (%.1f M points/s)" % (xyz.shape[0], off.shape[0] - 1, t, xyz.shape[0] / t / 1e3))
    m = 30000
    t0 = time.perf_counter()
    orc.pointgroup_clusters(xyz[:m], sem[:m], 0.03, 50)
    tc = time.perf_counter() - t0
    print("oracle (KD-tree + python BFS, 1 core) on the first %d points: %.0f ms (%.3f M points/s)" % (m, tc * 1e3, m / tc / 1e6))


def coarse():
    """coarse-level layer shapes (L2..L4 of an 8-scene batch)"""
    coords, feats, labels = make_batch(list(range(8)), n_target=150000, shift_seed=0)
    c = torch.from_numpy(coords).to(DEV)
    x = ME.SparseTensor(torch.zeros(coords.shape[0], 3, device=DEV), c)
    m = x.coordinate_manager
    k = x.coordinate_map_key
    for lvl in range(1, 5):
        k = m.stride(k, 2)
        n = m.size(k)
        km = m.kernel_map_handle(k, k, 3)
        M = km.export()[0].shape[0]
        for cin, cout in {1: [(32, 32), (96, 96)], 2: [(64, 64), (128, 128)], 3: [(128, 128), (256, 256)], 4: [(256, 256)]}[lvl]:
            f = torch.randn(n, cin, device=DEV).bfloat16()
            g = torch.randn(n, cout, device=DEV).bfloat16()
            w = torch.randn(27, cin, cout, device=DEV) * 0.05
            tf = timeit(lambda: km.conv_forward(f, w, None, False), 10, 3)
            tw = timeit(lambda: km.conv_wgrad(f, g, False), 10, 3)
            flop = 2.0 * M * cin * cout
            print("L%d rows %7d pairs %8d  %3d->%3d  fwd %.3f ms (%.0f TF)  wgrad %.3f ms (%.0f TF)" % (
                lvl, n, M, cin, cout, tf, flop / tf / 1e9, tw, flop / tw / 1e9))


def wgrad():
    """every weight-gradient shape of a Res16UNet34C step on the maps of the 8-scene batch (bf16)"""
    coords, feats, labels = make_batch(list(range(8)), n_target=150000, shift_seed=0)
    c = torch.from_numpy(coords).to(DEV)
    x = ME.SparseTensor(torch.zeros(coords.shape[0], 3, device=DEV), c)
    m = x.coordinate_manager
    keys = [x.coordinate_map_key]
    for lvl in range(4):
        keys.append(m.stride(keys[-1], 2))
    # (level, kernel, cin, cout, transposed, count per step)
    if os.environ.get("LGS_WGRAD_DBG"):
        shapes_sel = [(0, 3, 96, 96, 0, 3), (0, 3, 128, 96, 0, 1), (1, 3, 96, 96, 0, 3)]
    shapes = [(0, 3, 3, 32, 0, 1), (0, 3, 128, 96, 0, 1), (0, 3, 96, 96, 0, 3), (0, 2, 32, 32, 0, 1), (0, 2, 96, 96, 1, 1),
              (1, 3, 32, 32, 0, 4), (1, 3, 128, 96, 0, 1), (1, 3, 96, 96, 0, 3), (1, 2, 32, 32, 0, 1), (1, 2, 128, 96, 1, 1),
              (2, 3, 32, 64, 0, 1), (2, 3, 64, 64, 0, 5), (2, 3, 192, 128, 0, 1), (2, 3, 128, 128, 0, 3), (2, 2, 64, 64, 0, 1), (2, 2, 256, 128, 1, 1),
              (3, 3, 64, 128, 0, 1), (3, 3, 128, 128, 0, 7), (3, 3, 384, 256, 0, 1), (3, 3, 256, 256, 0, 3), (3, 2, 128, 128, 0, 1), (3, 2, 256, 256, 1, 1),
              (4, 3, 128, 256, 0, 1), (4, 3, 256, 256, 0, 11)]
    tot = 0.0
    if os.environ.get("LGS_WGRAD_DBG"):
        shapes = shapes_sel
    for lvl, ks, cin, cout, tr, cnt in shapes:
        if ks == 3:
            km = m.kernel_map_handle(keys[lvl], keys[lvl], 3)
            n_in = n_out = m.size(keys[lvl])
        else:
            km = m.kernel_map_handle(keys[lvl], keys[lvl + 1], 2)
            n_in, n_out = (m.size(keys[lvl + 1]), m.size(keys[lvl])) if tr else (m.size(keys[lvl]), m.size(keys[lvl + 1]))
        M = km.export()[0].shape[0]
        f = torch.randn(n_in, cin, device=DEV).bfloat16()
        g = torch.randn(n_out, cout, device=DEV).bfloat16()
        tw = timeit(lambda: km.conv_wgrad(f, g, bool(tr)), 10, 3)
        flop = 2.0 * M * cin * cout
        byts = M * (cin + cout) * 2 + 8 * M
        tot += tw * cnt
        print("L%d k%d%s %3d->%3d rows %7d pairs %8d x%-2d  wgrad %.3f ms (%.0f TF, %.2f TB/s alg)" % (
            lvl, ks, "T" if tr else " ", cin, cout, n_out, M, cnt, tw, flop / tw / 1e9, byts / tw / 1e9))
    print("sum over a step (stand-alone launches): %.2f ms" % tot)


def wide():
    """the wide-channel launches of the CLIP pretrain model (Res16UNet34D, BASELINE configs[2]) on the 8-scene maps"""
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    coords, feats, labels = make_batch(list(range(B)), n_target=150000, shift_seed=0)
    c = torch.from_numpy(coords).to(DEV)
    x = ME.SparseTensor(torch.zeros(coords.shape[0], 3, device=DEV), c)
    m = x.coordinate_manager
    keys = [x.coordinate_map_key]
    for lvl in range(2):
        keys.append(m.stride(keys[-1], 2))
    for lvl, cin, cout in ((0, 512, 512), (0, 640, 512), (1, 256, 256), (1, 512, 256), (2, 256, 256)):
        km = m.kernel_map_handle(keys[lvl], keys[lvl], 3)
        n = m.size(keys[lvl])
        M = km.export()[0].shape[0]
        f = torch.randn(n, cin, device=DEV).bfloat16()
        g = torch.randn(n, cout, device=DEV).bfloat16()
        w = torch.randn(27, cin, cout, device=DEV) * 0.02
        tf = timeit(lambda: km.conv_forward(f, w, None, False), 3, 1)
        td = timeit(lambda: km.conv_dgrad(g, w, False), 3, 1)
        tw = timeit(lambda: km.conv_wgrad(f, g, False), 3, 1)
        flop = 2.0 * M * cin * cout
        print("L%d 3^3 %3d->%3d rows %7d pairs %8d  fwd %.3f ms (%.0f TF = %.1f %% of 2.5 PF)  dgrad %.3f ms (%.0f TF)  wgrad %.3f ms (%.0f TF)" % (
            lvl, cin, cout, n, M, tf, flop / tf / 1e9, flop / tf / 1e9 / 25, td, flop / td / 1e9, tw, flop / tw / 1e9))
    # the other wide launches of Res16UNet34D at level 0: 1x1 downsample 544 -> 512, transposed 2^3 256 -> 512 (level 1 -> 0)
    n0 = m.size(keys[0])
    km1 = m.kernel_map_handle(keys[0], keys[0], 1)
    f = torch.randn(n0, 544, device=DEV).bfloat16()
    g = torch.randn(n0, 512, device=DEV).bfloat16()
    w = torch.randn(1, 544, 512, device=DEV) * 0.02
    tf = timeit(lambda: km1.conv_forward(f, w, None, False), 3, 1)
    td = timeit(lambda: km1.conv_dgrad(g, w, False), 3, 1)
    print("L0 1x1 544->512 rows %7d  fwd %.3f ms (%.0f TF)  dgrad %.3f ms (%.0f TF)" % (n0, tf, 2.0 * n0 * 544 * 512 / tf / 1e9, td, 2.0 * n0 * 544 * 512 / td / 1e9))
    km2 = m.kernel_map_handle(keys[0], keys[1], 2)
    n1 = m.size(keys[1])
    f = torch.randn(n1, 256, device=DEV).bfloat16()
    w = torch.randn(8, 256, 512, device=DEV) * 0.02
    tf = timeit(lambda: km2.conv_forward(f, w, None, True), 3, 1)
    td = timeit(lambda: km2.conv_dgrad(g, w, True), 3, 1)
    print("L1->L0 transposed 2^3 256->512 rows %7d  fwd %.3f ms (%.0f TF)  dgrad %.3f ms (%.0f TF)" % (n0, tf, 2.0 * n0 * 256 * 512 / tf / 1e9, td, 2.0 * n0 * 256 * 512 / td / 1e9))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        wide()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        wgrad()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coarse":
        coarse()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cluster":
        cluster()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "insseg":
        insseg()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "quantize":
        quantize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "clip":
        clip()
        sys.exit(0)
    main()
