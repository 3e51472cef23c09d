import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import MinkowskiEngine as ME
be = ME.get_backend()
DEV = "cuda:0"
for c, na, n in ((512, 200, 257), (64, 20, 130), (32, 64, 4099)):
    g = torch.Generator().manual_seed(1)
    F = torch.randn(n, c, generator=g).bfloat16().float()
    T = torch.randn(na, c, generator=g)
    lab = torch.randint(0, na, (n,), generator=g)
    neg = torch.randint(0, na, (n, 3), generator=g)
    d_pos, d_neg, pred, saved, sim = be.clip_loss_forward(F.to(DEV).bfloat16(), T.to(DEV), lab.to(DEV), neg.to(DEV), -1, want_sim=True)
    inv = saved[4].cpu()
    r = inv * F.norm(dim=1)
    print(c, na, n, "inv*|f| min/max", float(r.min()), float(r.max()))
    fn = torch.nn.functional.normalize(F.double(), dim=1); tn = torch.nn.functional.normalize(T.double(), dim=1)
    sr = fn @ tn.t()
    print("  sim err max", float((sim.cpu().double() - sr).abs().max()), "dpos err", float((d_pos.cpu().double() - (1 - sr.gather(1, lab[:, None]).squeeze(1))).abs().max()))
    s2, inv2 = be.clip_similarity(F.to(DEV).bfloat16(), T.to(DEV))
    print("  dense path sim err", float((s2.cpu().double() - sr).abs().max()), "fused-vs-dense", float((s2 - sim).abs().max()))
