#!/usr/bin/env python
"""bench.py -- voxels/sec of the Res16UNet34C training step (forward + loss + backward + gradient
all-reduce + SGD step) on synthetic ScanNet200-shaped scenes, one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 10 --warmup 3

A "step" is one pass of the hot path over one batch of `--scenes` synthetic scenes per GPU (~150k voxels
each @2cm, BASELINE.json configs[1]); inputs are resident in HBM when the timed region starts; the step
includes the trainer's per-step coordinate shift, SparseTensor construction (coordinate hashing + all kernel
maps), forward, cross-entropy over 200 classes, backward, RCCL gradient all-reduce (N>1) and the SGD update.
Scenes shard over ranks with no data-path collective ("weak" scaling: per-GPU work fixed).

Rank 0 prints ONE JSON line; besides the driver's contract it carries
  roofline     -- the dominant kernel family (k_conv_gather: sparse-conv forward/dgrad implicit GEMM),
                  algorithmic bytes of every launch of one step (SURVEY 8d byte model) / their HIP-event
                  durations, vs the 8 TB/s HBM peak; plus the whole-step B_alg fraction
  cpu_baseline -- the oracle's BLAS gather-GEMM-scatter restatement of the same model on the host cores,
                  on a bounded sample (it is a restatement of MinkowskiEngine's CPU algorithm, not ME itself)
"""
import argparse
import json
import os
import sys
import time

# before the HIP runtime starts (see languagegroundedsemseg_amd/__init__.py); ranks SHARING one GPU (--same-device dry runs) keep
# the runtime's 4: 2 x 8 queues oversubscribe the device's hardware queues (two gloo ranks on one GPU: 136 vs 520 ms per step)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if "--same-device" in sys.argv else "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import MinkowskiEngine as ME  # noqa: E402
from languagegroundedsemseg_amd import models  # noqa: E402
from languagegroundedsemseg_amd.me import block as _block  # noqa: E402
from languagegroundedsemseg_amd.me import deferred as _deferred  # noqa: E402
from languagegroundedsemseg_amd.ddp import BucketedDDP, FlatSGD  # noqa: E402
from languagegroundedsemseg_amd.losses import fused_cross_entropy  # noqa: E402
from languagegroundedsemseg_amd.synthetic import make_batch  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md; 6.29 TB/s measured float4 copy)


class Cfg:
    bn_momentum = 0.02
    conv1_kernel_size = 3
    dilations = [1, 1, 1, 1]


# ----------------------------------------------------------------------------------------------- byte model
class ConvLog:
    """Wraps HipKernelMap conv entry points to log (kind, M, n_out, cin, cout, K, HIP events) per launch.
    mode "all": every launch (the one instrumented DISCOVERY step: ~250 event pairs stretch that step, so its numbers
    only rank the shapes); mode "only": just the launches of one shape key (a handful of event pairs per step, used
    INSIDE the timed steps -- the reported roofline comes from these)."""

    def __init__(self):
        self.rows = []
        self.mode = None          # None | "all" | "only"
        self.only_key = None
        self._patched = False

    @staticmethod
    def key_of(kind, K, cin, cout, n_out):
        return "conv_gather K=%d %d->%d rows=%d" % (K, cin, cout, n_out)

    def patch(self):
        if self._patched:
            return
        from languagegroundedsemseg_amd.me import backend_hip as bh
        log = self
        pairs_cache = {}

        def n_pairs(km):
            key = (id(km.mgr), km.in_key, km.out_key, km.ks)
            if key not in pairs_cache:
                pairs_cache[key] = int(km.export()[0].shape[0])
            return pairs_cache[key]

        def wrap(name, kind):
            orig = getattr(bh.HipKernelMap, name)

            def f(self, *a, **k):
                if log.mode is None:
                    return orig(self, *a, **k)
                if log.mode == "roctx":      # profiling runs (tools/collect_profiles.sh): name the launches of this call by shape
                    if kind == "wgrad":
                        nm = "lgs wgrad K=%d %d->%d rows=%d" % (self.K, a[0].shape[1], a[1].shape[1], a[1].shape[0])
                    elif kind == "fwd":
                        nm = "lgs conv_fwd K=%d %d->%d rows=%d" % (self.K, a[0].shape[1], a[1].shape[-1], self._rows(a[3])[1])
                    else:
                        nm = "lgs conv_dgrad K=%d %d->%d rows=%d" % (self.K, a[0].shape[1], a[1].shape[-2], self._rows(a[2])[0])
                    log.roctx.roctxRangePushA(nm.encode())
                    try:
                        return orig(self, *a, **k)
                    finally:
                        log.roctx.roctxRangePop()
                if kind == "wgrad":
                    cin, cout, n_out = a[0].shape[1], a[1].shape[1], a[1].shape[0]
                elif kind == "fwd":
                    cin, cout = a[0].shape[1], a[1].shape[-1]
                    n_in, n_out = self._rows(a[3])
                else:  # dgrad: "output" side is the op's input
                    cin, cout = a[0].shape[1], a[1].shape[-2]
                    n_out, _ = self._rows(a[2])
                key = ConvLog.key_of(kind, self.K, cin, cout, n_out)
                if log.mode == "only" and (kind == "wgrad" or key != log.only_key):
                    return orig(self, *a, **k)
                M = n_pairs(self) if log.mode == "all" else log.M_of.get(key, 0)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st = k.get("stream")          # weight gradients are enqueued on the side stream explicitly
                s.record(st) if st is not None else s.record()
                out = orig(self, *a, **k)
                e.record(st) if st is not None else e.record()
                log.rows.append(dict(kind=kind, M=M, n_out=n_out, cin=cin, cout=cout, K=self.K, ev=(s, e),
                                     e=a[0].element_size(), key=key))
                return out
            setattr(bh.HipKernelMap, name, f)
        wrap("conv_forward", "fwd")
        wrap("conv_dgrad", "dgrad")
        wrap("conv_wgrad", "wgrad")
        self.M_of = {}
        self._patched = True
        # residual blocks are ONE engine call per direction (lgs_block_forward / lgs_block_backward): their conv launches cannot
        # be bracketed from here.  A block whose launches are wanted is enqueued call by call (same launches, bit-identical):
        # every block in the discovery step and under --roctx, only the blocks containing the dominant shape in the timed steps.
        def veto(rows, cin, planes):
            if log.mode in ("all", "roctx"):
                return True
            if log.mode == "only":
                mine = {ConvLog.key_of("", 27, cin, planes, rows), ConvLog.key_of("", 27, planes, planes, rows),
                        ConvLog.key_of("", 27, planes, cin, rows), ConvLog.key_of("", 1, cin, planes, rows),
                        ConvLog.key_of("", 1, planes, cin, rows)}
                return log.only_key in mine
            return False
        _block._BLOCK_C_VETO = veto

    def summarize(self):
        """-> (family totals of k_conv_gather launches, wgrad totals, per-shape groups sorted by time)"""
        torch.cuda.synchronize()
        fam = dict(bytes=0.0, ms=0.0, n=0)      # k_conv_gather family = fwd + dgrad launches
        wg = dict(bytes=0.0, ms=0.0, n=0)
        groups = {}
        for r in self.rows:
            t = r["ev"][0].elapsed_time(r["ev"][1])
            e, M, K = r["e"], r["M"], r["K"]
            idx = 0 if K == 1 else 8 * M
            if r["kind"] == "wgrad":
                b = M * (r["cin"] + r["cout"]) * e + idx + K * r["cin"] * r["cout"] * 4
                wg["bytes"] += b; wg["ms"] += t; wg["n"] += 1
            else:
                b = M * r["cin"] * e + r["n_out"] * r["cout"] * e + idx + K * r["cin"] * r["cout"] * e
                fam["bytes"] += b; fam["ms"] += t; fam["n"] += 1
                self.M_of[r["key"]] = M
                g = groups.setdefault(r["key"], dict(bytes=0.0, ms=0.0, n=0, flop=0.0))
                g["bytes"] += b; g["ms"] += t; g["n"] += 1; g["flop"] += 2.0 * M * r["cin"] * r["cout"]
        top = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
        return fam, wg, top


# ----------------------------------------------------------------------------------------------- the step
def build(device, dtype, n_classes=200, model_name="Res16UNet34C"):
    torch.manual_seed(42)  # config.py:276
    model = models.load_model(model_name)(3, n_classes, Cfg()).to(device).train()
    return model


_DATA_STREAM = {}
_PHASES = []      # (tag, HIP event on the compute stream, host time) -- five marks per step, drained by phase_summary()


def _mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    _PHASES.append((tag, e, time.perf_counter()))


def phase_summary(steps, origin=None):
    """per-step milliseconds between the marks of the last `steps` steps: on the compute stream (HIP events) and on the
    host (enqueue time).  `finalize` = joining the weight-gradient stream + (N > 1) the exposed part of the bucket
    all-reduces; call after torch.cuda.synchronize()."""
    ev = _PHASES[-5 * steps:]
    acc, host = {}, {}
    for j in range(steps):
        blk = ev[5 * j:5 * j + 5]
        for a, b in zip(blk[:-1], blk[1:]):
            acc[b[0]] = acc.get(b[0], 0.0) + a[1].elapsed_time(b[1])
            host[b[0]] = host.get(b[0], 0.0) + (b[2] - a[2]) * 1e3
    out = {"stream_ms": {k: v / steps for k, v in acc.items()}, "host_enqueue_ms": {k: v / steps for k, v in host.items()}}
    if origin is not None:
        # how far the HOST runs ahead of the compute stream at each mark: (time the stream reached the mark) - (time the host
        # enqueued it), both from the synchronised start of the timed region.  ~0 at `start` = the host is not ahead at the
        # head of a step (the stream waits for the host); growing over the steps = nothing throttles the host
        ev0, t0 = origin
        lead = {}
        for j in range(steps):
            for tag, e, t in ev[5 * j:5 * j + 5]:
                lead.setdefault(tag, []).append(ev0.elapsed_time(e) - (t - t0) * 1e3)
        out["host_lead_ms"] = {k: {"avg": sum(v) / len(v), "first_step": v[0], "last_step": v[-1]} for k, v in lead.items()}
    del _PHASES[:]
    return out


def train_step(model, ddp, opt, coords, feats, labels, dtype, step_idx, shift=True, ctx=None):
    """One training step.  The input side (the trainer's per-step coordinate shift, the feature cast and the
    SparseTensor construction = coordinate insert) runs on a separate "data" stream, like a DataLoader's copy
    stream: the engine builds all coordinate / kernel maps on the manager's own stream, so the maps of step t+1 are
    constructed while step t's backward is still running on the compute stream.
    ctx: None = cross-entropy fine-tune (configs[1]); {"kind": "clip", crit, anchors} = CLIP-contrastive pretrain
    (configs[2], pl_RepresentationTrainer.py:168-264); {"kind": "insseg", inst, centers} = instance-segmentation step
    (configs[4], downstream/insseg/lib/pl_Trainer.py:245-321: CE + offset-L1 + direction losses)."""
    main = torch.cuda.current_stream()
    dev = coords.device
    ds = _DATA_STREAM.setdefault(dev.index, torch.cuda.Stream(device=dev))
    with torch.cuda.stream(ds):
        c = coords
        if shift:  # pl_BaselineTrainer.py:294: random integer shift, same for the whole batch
            g = torch.Generator().manual_seed(1000 + step_idx)
            sh = (torch.rand(3, generator=g) * 100).to(torch.int32).tolist()
            c = coords.clone()
            for d in range(3):          # scalar adds: no host->device copy (and its implicit sync)
                c[:, 1 + d] += sh[d]
        f = feats.to(dtype)
        sinput = ME.SparseTensor(f, c)                                 # coordinate hash; kernel maps follow lazily
    main.wait_stream(ds)
    for t in (c, f, sinput.F):
        t.record_stream(main)
    ddp.zero_grad()
    _mark("start")
    kind = ctx["kind"] if ctx else "ce"
    if kind == "clip":
        out = model(sinput)
        loss, _, _ = ctx["crit"](out.F, labels, ctx["anchors"])
    elif kind == "insseg":
        from languagegroundedsemseg_amd.losses import instance_offset_losses
        off, logits, _ = model(sinput)
        nl, dl = instance_offset_losses(off.F, c[:, 1:], ctx["centers"], ctx["inst"], 0.02)
        loss = fused_cross_entropy(logits.F, labels, ignore_index=-1) + nl + dl
    elif kind == "ce_balanced":
        # the fine-tune step as scripts/train_models.sh:37 runs it (--balanced_category_sampling True): CrossEntropyLoss(reduction=
        # 'none') -> sample_categories_for_balancing -> masked mean over ALL points (pl_BaselineTrainer.py:94,350-356,
        # lib/losses/utils.py:13-77); split statistics stay on the device (what the trainer's head / common / tail meters consume)
        from languagegroundedsemseg_amd.losses import sample_categories_for_balancing
        logits, _ = model(sinput)
        rows = fused_cross_entropy(logits.F, labels, ignore_index=-1, reduction="none")
        loss, ctx["split_stats"], _ = sample_categories_for_balancing(rows, labels, ctx["foc"], ctx["head_ratio"], ctx["common_ratio"],
                                                                       ignore_label=-1, split="stats")
    else:
        logits, _ = model(sinput)
        loss = fused_cross_entropy(logits.F, labels, ignore_index=-1)
    _mark("forward")
    loss.backward()
    _mark("backward")       # compute stream done with backward; the weight-gradient stream may still be busy
    ddp.finalize()
    _mark("finalize")       # joined the weight-gradient stream (+ gradient all-reduce when N > 1)
    opt.step()
    _mark("optimizer")
    return loss


def timed_steps(model, ddp, opt, coords, feats, labels, dtype, steps, warmup, ctx=None, base=20000):
    """`warmup` untimed + `steps` timed steps of a secondary workload -> (ms per step, phases)"""
    for i in range(warmup):
        train_step(model, ddp, opt, coords, feats, labels, dtype, base + i, ctx=ctx)
    torch.cuda.synchronize()
    del _PHASES[:]
    t0 = time.perf_counter()
    for i in range(steps):
        train_step(model, ddp, opt, coords, feats, labels, dtype, base + 100 + i, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3, phase_summary(steps)


def cpu_baseline(seconds_budget=30.0, model_name="Res16UNet34C", voxels=150000, voxel=0.02, loss="ce", max_steps=3):
    """Oracle ("port"): MinkowskiEngine-CPU-style gather -> BLAS GEMM -> scatter restated with torch CPU ops, same
    model, ONE synthetic scene (default: 2 cm, ~150k voxels = BASELINE.md section 2, config 2 -- the scene unit the GPU runs eight
    of per step), host cores, fwd + loss + bwd.  Bounded: one warm-up step, then timed steps until the budget is spent.
    loss="clip": the text-anchor contrastive loss of configs[2] (dense similarity + gathers on the CPU)."""
    from oracle.backend import OracleBackend
    prev = ME.set_backend(OracleBackend("torch"))
    try:
        cores = min(host_cores(), 16)  # more threads only add OpenMP contention to the small per-offset GEMMs
        torch.set_num_threads(cores)
        coords, feats, labels = make_batch([0], voxel=voxel, n_target=voxels)
        torch.manual_seed(42)
        model = models.load_model(model_name)(3, 200, Cfg()).train()
        c, f, l = torch.from_numpy(coords), torch.from_numpy(feats), torch.from_numpy(labels)
        crit = anchors = None
        if loss == "clip":
            from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
            from languagegroundedsemseg_amd.synthetic import text_anchors
            model.representation_only(True)
            crit = ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3)
            anchors = torch.from_numpy(text_anchors(200, model.PLANES[7]))

        def one():
            for p in model.parameters():
                p.grad = None
            x = ME.SparseTensor(f, c)
            if crit is not None:
                out = crit(model(x).F, l, anchors)[0]
            else:
                logits, _ = model(x)
                out = torch.nn.functional.cross_entropy(logits.F, l, ignore_index=-1)
            out.backward()
        t_all = time.perf_counter()
        one()  # warm-up (builds nothing persistent: maps are per step, as in the reference)
        times = []
        while len(times) < max_steps and (not times or (time.perf_counter() - t_all) < seconds_budget):
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
        best = min(times)
        return {"value": coords.shape[0] / best, "unit": "voxels/s", "cores": cores, "kind": "port",
                "sample": "%s fwd+%s loss+bwd, 1 synthetic scene @%gcm (%d voxels), fp32, best of %d after 1 warm-up (%.1f s of CPU "
                          "work); oracle BLAS gather-GEMM-scatter restatement of ME's CPU algorithm (not ME itself)" % (
                              model_name, loss, voxel * 100, coords.shape[0], len(times), time.perf_counter() - t_all)}
    finally:
        ME.set_backend(prev)


def cpu_baselines_other_configs():
    """BASELINE.md section 2's other two CPU rows, bounded (details file only; the line's `cpu_baseline` is configs[1]'s):
    configs[0] Res16UNet14A @5 cm, one scene -- the reference's own CPU-runnable case; configs[2] Res16UNet34D + contrastive
    text-anchor loss on a 40k-voxel 2 cm scene (a full 150k-voxel scene of the 512-channel decoder is ~1 min per CPU step)."""
    return {"configs0_res16unet14a_5cm": cpu_baseline(8.0, "Res16UNet14A", voxels=25000, voxel=0.05, max_steps=3),
            "configs2_res16unet34d_clip_sample": cpu_baseline(20.0, "Res16UNet34D", voxels=40000, voxel=0.02, loss="clip", max_steps=2)}


def pmc_traffic(dom_key, args):
    """HBM bytes per launch of the dominant kernel.  STATIC: read from the committed PMC passes of this round
    (profiles/rNN_pmc_traffic.json, newest round first: FETCH_SIZE / WRITE_SIZE collected with rocprofv3 --pmc in separate passes, read side
    doubled as MI355X_MICROARCH.md prescribes for gfx950); only valid for the default workload they were collected on, null
    otherwise.  It is not measured inside this run (PMC collection needs the profiler)."""
    wide = "K=27 512->512" in dom_key                 # the CLIP workload's dominant shape (k_conv_wide)
    import glob
    suffix = "_pmc_traffic_wide.json" if wide else "_pmc_traffic.json"
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + suffix)), reverse=True)     # newest round first
    if not found:
        return None, None
    path = found[0]
    name = os.path.basename(path)
    if not (args.scenes == 8 and args.voxels == 150000 and args.dtype == "bf16" and ("K=27 96->96" in dom_key or wide)):
        return None, None
    try:
        return json.load(open(path))["traffic_bytes"], "static: profiles/%s (rocprofv3 --pmc passes of this workload)" % name
    except Exception:
        return None, None


def single_scene_line(model, ddp, opt, dtype, device, args, ctx=None, steps=40, warmup=8):
    """Secondary line: ONE ~150k-voxel scene per step (the unit SURVEY 8d tabulates).  With ~720 launches on the critical
    path the step is launch / latency bound at this size; reported so the 8-scene headline is not read as per-scene."""
    c_np, f_np, l_np = make_batch([1000], voxel=0.02, n_target=args.voxels)
    c, f, l = torch.from_numpy(c_np).to(device), torch.from_numpy(f_np).to(device), torch.from_numpy(l_np).to(device)
    for i in range(warmup):
        train_step(model, ddp, opt, c, f, l, dtype, 5000 + i, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        train_step(model, ddp, opt, c, f, l, dtype, 6000 + i, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"scenes_per_step": 1, "voxels": int(c.shape[0]), "ms_per_step": dt / steps * 1e3, "value": c.shape[0] * steps / dt,
            "unit": "voxels/s", "steps": steps}


def clip_mfma_report(out_dim, n, dtype, device, iters=10):
    """MFMA sub-report of north_star for the one dense contraction (per-voxel features x 200 text anchors) as the fused
    CLIP-loss kernel runs it: achieved TFLOP/s vs the 2.5 PF bf16 peak and its bytes (features read once) vs 8 TB/s."""
    be = ME.get_backend()
    f = torch.randn(n, out_dim, device=device).to(dtype)
    t = torch.randn(200, out_dim, device=device)
    lab = torch.randint(-1, 200, (n,), device=device)
    neg = torch.randint(0, 200, (n, 3), device=device)
    for _ in range(3):
        be.clip_loss_forward(f, t, lab, neg, -1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        be.clip_loss_forward(f, t, lab, neg, -1)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    flop = 2.0 * n * out_dim * 200
    byts = n * out_dim * f.element_size() + n * (3 * 4 + 8 + 4 * 8)
    peak = 2.5e15 if dtype == torch.bfloat16 else 157.3e12
    return {"kernel": "k_conv_gather<EPI=1> fused CLIP loss forward (normalize + [N,%d]x[%d,200] + gathers + argmax)" % (out_dim, out_dim),
            "ms": ms, "achieved": flop / (ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / peak,
            "hbm": {"achieved": byts / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": byts / (ms * 1e-3) / HBM_PEAK},
            "note": "arithmetic intensity ~%d flop/B: the contraction is bandwidth-limited at this K (SURVEY 8d)" % int(flop / byts)}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %8.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def workload_text(workload, model_name):
    if workload == "ce":
        return ("%s 2cm ScanNet200-shaped synthetic scenes, cross-entropy fine-tune step (configs[1]): SparseTensor build + "
                "fwd + CE(200) + bwd + grad all-reduce + SGD" % model_name)
    if workload.startswith("ce_balanced"):
        return ("%s fine-tune step with --balanced_category_sampling True (scripts/train_models.sh:37): per-point CE + "
                "sample_categories_for_balancing (%s) + bwd + SGD" % (
                    model_name, "head / common ratios 0.5" if workload.endswith("sampled") else "ratios -1 = config defaults"))
    if workload == "clip":
        return ("%s 2cm synthetic scenes, CLIP-contrastive pretrain step (configs[2]): SparseTensor build + fwd + text-anchor "
                "contrastive loss (200 anchors, MFMA contraction) + bwd + grad all-reduce + SGD" % model_name)
    return ("%s 2cm synthetic scenes, instance-segmentation step (configs[4]%s): SparseTensor build + fwd + CE(200) + offset-L1 + "
            "direction losses + bwd + SGD" % (model_name, ", frozen trunk" if workload == "insseg_frozen" else ""))


def make_ctx(workload, model_name, coords, device):
    """loss-side context of a workload (None = cross-entropy)"""
    if workload == "clip":
        from languagegroundedsemseg_amd.losses import ContrastiveLanguageLoss
        from languagegroundedsemseg_amd.synthetic import text_anchors
        dim = models.load_model(model_name).PLANES[7]
        return {"kind": "clip", "crit": ContrastiveLanguageLoss(num_labels=200, num_negative_samples=3),
                "anchors": torch.from_numpy(text_anchors(200, dim)).to(device)}
    if workload in ("ce_balanced", "ce_balanced_sampled"):
        # ScanNet200's split: 66 head / 68 common / 66 tail categories (lib/datasets/scannet.py:131-141, lib/constants); the synthetic
        # labels are uniform over the 200 classes, so which ids are "head" does not matter for the timing
        foc = torch.zeros(200, 3, dtype=torch.bool)
        foc[:66, 0], foc[66:134, 1], foc[134:, 2] = True, True, True
        sampled = workload.endswith("sampled")      # config.py:281-282 defaults are -1 / -1 (keep all); `sampled` draws half of head / common
        return {"kind": "ce_balanced", "foc": foc.to(device), "head_ratio": 0.5 if sampled else -1.0,
                "common_ratio": 0.5 if sampled else -1.0}
    if workload in ("insseg", "insseg_frozen"):
        g = torch.Generator().manual_seed(0)
        n = coords.shape[0]
        inst = torch.randint(-1, 30, (n,), generator=g).to(device)
        centers = coords[:, 1:].float() + (torch.randn(n, 3, generator=g) * 20).to(device)
        return {"kind": "insseg", "inst": inst, "centers": centers, "frozen": workload == "insseg_frozen"}
    return None


def make_trainer(model_name, dtype, device, world, args, ctx):
    model = build(device, dtype, model_name=model_name)
    if ctx and ctx["kind"] == "clip":
        model.representation_only(True)
    if ctx and ctx["kind"] == "insseg" and ctx["frozen"]:
        model.freeze_trunk(True)
    force = bool(getattr(args, "dp_world1", False)) and world == 1 and dist.is_initialized()
    if (world > 1 or force) and args.sync_bn:
        model = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
    ddp = BucketedDDP(model, bucket_mb=32.0, allreduce=args.allreduce, force_collectives=force)
    opt = FlatSGD(ddp, lr=1e-2, momentum=0.9, dampening=0.1, weight_decay=1e-4)  # same rule as lib/solvers.py's SGD
    return model, ddp, opt


SETTLE_ROUND = 5      # steps per warm-up round between two synchronisations


def _settled(rounds):
    """warm-up verdict from the per-step compute-stream times of the warm-up ROUNDS (each a run of steps between two
    synchronisations).  Inside a round the first two steps are slow (nothing of the first was prepared under a previous step, the
    second still catches up) and the last one fast (no next step builds its maps beside it), so a round is judged by the two steps
    before its last: they agree within 3 %, and their mean is within 2 % of the previous round's"""
    mids = [r[-3:-1] for r in rounds if len(r) >= 5]
    if len(mids) < 2:
        return False
    a, b = mids[-1]
    m1, m0 = 0.5 * (a + b), 0.5 * sum(mids[-2])
    return abs(a - b) <= 0.03 * min(a, b) and abs(m1 - m0) <= 0.02 * m0


def step_stats(ms):
    """median / p90 / min / max of the timed steps' compute-stream durations (the headline `value` is the wall-clock mean over
    exactly K steps, as the contract says; these say whether that mean carries a transient)"""
    v = sorted(ms)
    n = len(v)
    if not n:
        return None
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"median": med, "p90": v[min(n - 1, int(0.9 * n))], "min": v[0], "max": v[-1]}


def measure(model, ddp, opt, coords, feats, labels, dtype, steps, warmup, ctx, clog, world, want_roofline, base=0, settle=15):
    """`warmup` untimed plain steps, then -- still untimed -- up to `settle` more in rounds of five until two consecutive rounds
    agree (`_settled`) (a fresh box pays for allocator growth, code-object loading and clock ramps in its first steps: round 5's
    driver run had six 35 - 56 ms steps inside the timed region); then exactly `steps` timed steps bracketed by barrier +
    synchronize on both sides.  The fully instrumented DISCOVERY step and the SAMPLING steps of the roofline run AFTER the timed
    region: nothing before or inside it is bracketed, enqueued call by call or allocated differently from production.
    -> dict(dt, loss, disc, step_ms, phases, warmup_ms, warmup_extra)"""
    disc = None
    if clog is not None:
        clog.rows, clog.mode, clog.only_key = [], None, None

    def plain(n, first):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            train_step(model, ddp, opt, coords, feats, labels, dtype, first + i, ctx=ctx)
            ev[i + 1].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]

    rounds = [plain(warmup, base)] if warmup > 0 else []
    extra = 0
    while settle > 0 and extra < settle and warmup > 0:
        ok = _settled(rounds)
        if world > 1:      # every rank runs the same number of steps (the steps contain collectives)
            flag = torch.tensor([0.0 if ok else 1.0], device=coords.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            ok = flag.item() == 0.0
        if ok:
            break
        rounds.append(plain(SETTLE_ROUND, base + warmup + extra))
        extra += SETTLE_ROUND
    warm_ms = [m for r in rounds for m in r]
    if ddp.timing is not None:
        ddp.timing_summary(1)                   # drop the warm-up's collective events
    del _PHASES[:]
    log("warmup done (%d + %d steps): %s ms" % (warmup, extra, " ".join("%.1f" % m for m in warm_ms)))
    first = base + warmup + extra
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    for i in range(steps):
        loss = train_step(model, ddp, opt, coords, feats, labels, dtype, first + i, ctx=ctx)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"dt": dt, "loss": loss, "disc": None, "phases": phase_summary(steps, origin=(marks[0], t0)),
           "step_ms": [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)], "warmup_ms": warm_ms, "warmup_extra": extra}
    if ddp.timing is not None:
        out["ddp"] = ddp.timing_summary(steps)
    if want_roofline:
        # DISCOVERY step (every rank runs it: it contains the step's collectives): every conv launch is bracketed by HIP events to
        # rank the launch shapes and to evaluate the byte model on the real maps; this stretches the step, so its durations are
        # only used to pick the dominant shape
        if clog is not None:
            clog.rows, clog.mode = [], "all"
        train_step(model, ddp, opt, coords, feats, labels, dtype, first + steps, ctx=ctx)
        if clog is not None:
            fam, wg, top = clog.summarize()
            disc = dict(fam=fam, wg=wg, top=top, rows=list(clog.rows))
            # SAMPLING pass: a few more steps in which only the dominant shape's launches are bracketed by HIP events on their
            # stream -- its residual blocks are enqueued call by call for that (same launches, bit-identical), everything else
            # runs as in the timed steps
            clog.rows, clog.mode, clog.only_key = [], "only", top[0][0]
        n_samp = max(2, min(steps, 6))
        for i in range(n_samp):
            train_step(model, ddp, opt, coords, feats, labels, dtype, first + steps + 1 + i, ctx=ctx)
        torch.cuda.synchronize()
        if clog is not None:
            clog.mode = None
        out["disc"] = disc
        out["sample_steps"] = n_samp
        if ddp.timing is not None:
            ddp.timing_summary(1)
        del _PHASES[:]
    return out


def roofline_report(clog, disc, dtype_name, workload, steps, ms_per_step, n_vox, traffic_pair, sample_steps=0):
    """the `roofline` object of the JSON line from a ConvLog's discovery step + its in-step samples of the dominant shape"""
    clog.mode = None
    fam, wg, top = disc["fam"], disc["wg"], disc["top"]
    e = 2 if dtype_name == "bf16" else 4
    # BN byte model: 3 N C e fwd + 5 N C e bwd per norm layer; every BN follows exactly one conv forward
    # launch with the same (n_out, cout), except the classifier `final` (fine-tune workload)
    fwd_rows = [r for r in disc["rows"] if r["kind"] == "fwd"]
    bn_bytes = sum(8.0 * r["n_out"] * r["cout"] * e for r in (fwd_rows[:-1] if workload == "ce" else fwd_rows))
    b_alg_step = fam["bytes"] + wg["bytes"] + bn_bytes
    dom_key, dom_disc = top[0]
    # the dominant shape's launches sampled INSIDE the timed steps (un-instrumented otherwise)
    samp = [r for r in clog.rows if r["key"] == dom_key]
    samp_ms = [r["ev"][0].elapsed_time(r["ev"][1]) for r in samp]
    n_s = max(len(samp_ms), 1)
    avg_ms = sum(samp_ms) / n_s
    alg_bytes = dom_disc["bytes"] / max(dom_disc["n"], 1)
    flop = dom_disc["flop"] / max(dom_disc["n"], 1)
    achieved = alg_bytes / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
    traffic, traffic_src = traffic_pair
    fam_ach = fam["bytes"] / (fam["ms"] * 1e-3) if fam["ms"] > 0 else 0.0
    mfma_peak = 2.5e15 if dtype_name == "bf16" else 157.3e12
    # which roof bounds the dominant launch: its arithmetic intensity on the algorithmic bytes against the ridge point of the dtype
    # (2.5 PF / 8 TB/s = 312 flop/B for bf16); `achieved / peak / unit / frac` are quoted on THAT roof, both fractions stay in the object
    ai = flop / alg_bytes if alg_bytes else 0.0
    mfma_bound = ai > mfma_peak / HBM_PEAK
    tfl = flop / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    head = ({"bound": "mfma", "achieved": tfl, "peak": mfma_peak / 1e12, "unit": "TFLOP/s", "frac": tfl * 1e12 / mfma_peak}
            if mfma_bound else
            {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK})
    head.update({
        "arithmetic_intensity_flop_per_byte": ai, "hbm_GBps_on_alg_bytes": achieved / 1e9, "hbm_frac": achieved / HBM_PEAK,
        "kernel": ("k_conv_wide (2-D blocked wide-channel sparse conv forward/dgrad)" if "512->" in dom_key or "->512" in dom_key
                   else "k_conv_gather (sparse-conv forward/dgrad implicit GEMM)") + ", dominant launch shape: " + dom_key,
        "traffic": traffic, "traffic_source": traffic_src,
        # launches of this shape in ONE step (forward + dgrad launches; the 3^3 96->96 shape of Res16UNet34C: 4 forward + 5
        # backward ... counted from the discovery step, the same number DESIGN.md quotes)
        "launches_sampled": len(samp_ms), "launches_per_step": dom_disc["n"], "avg_launch_ms": avg_ms,
        "min_launch_ms": min(samp_ms) if samp_ms else None, "max_launch_ms": max(samp_ms) if samp_ms else None,
        "alg_bytes_per_launch": alg_bytes,
        "mfma_tflops_on_real_pairs": flop / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
        "mfma_frac_of_peak_on_real_pairs": flop / (avg_ms * 1e-3) / mfma_peak if avg_ms > 0 else 0.0,
        "measured": "HIP events on the launching stream around the %d launches of this shape inside %d SAMPLING steps run right after "
                    "the %d timed steps (same batch, same state; all other launches un-instrumented; the residual blocks that contain "
                    "this shape are enqueued call by call there instead of through lgs_block_forward / lgs_block_backward -- same "
                    "launches, bit-identical -- so that Python can bracket them).  The timed steps themselves carry no "
                    "instrumentation" % (len(samp_ms), sample_steps, steps),
        "discovery_step": {
            "note": "one fully instrumented step run AFTER the timed region (every conv launch bracketed: the step is stretched, "
                    "durations rank the shapes and feed the byte model only)",
            "family": {"kernel": "k_conv_gather, all %d launches of the step" % fam["n"], "achieved": fam_ach / 1e9,
                       "frac": fam_ach / HBM_PEAK, "total_ms": fam["ms"], "avg_launch_ms": fam["ms"] / max(fam["n"], 1)},
            "wgrad": {"kernel": "k_wgrad_ps / k_wgrad_bf16 + reduce, all %d launches (side stream)" % wg["n"],
                      "achieved": (wg["bytes"] / (wg["ms"] * 1e-3) / 1e9) if wg["ms"] > 0 else 0.0,
                      "frac": (wg["bytes"] / (wg["ms"] * 1e-3) / HBM_PEAK) if wg["ms"] > 0 else 0.0, "total_ms": wg["ms"]},
            "top_shapes": [{"shape": k, "launches": g["n"], "ms": g["ms"], "alg_GBps": g["bytes"] / (g["ms"] * 1e-3) / 1e9 if g["ms"] else 0.0,
                            "mfma_tflops": g["flop"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] else 0.0}
                           for k, g in top[:6]]},
        "step": {"b_alg_bytes": b_alg_step, "b_alg_per_voxel": b_alg_step / n_vox,
                 "frac_of_hbm_peak": b_alg_step / (ms_per_step * 1e-3) / HBM_PEAK},
    })
    return head


def secondary_block(workload, model_name, dtype, coords, feats, labels, device, args, clog, steps, warmup, note):
    """a few timed steps of another BASELINE configuration on the same resident batch -> dict for the JSON line"""
    import gc
    ctx = make_ctx(workload, model_name, coords, device)
    model, ddp, opt = make_trainer(model_name, dtype, device, 1, args, ctx)
    res = measure(model, ddp, opt, coords, feats, labels, dtype, steps, warmup, ctx, clog, 1, clog is not None, base=30000)
    n_vox = int(coords.shape[0])
    ms = res["dt"] / steps * 1e3
    dname = "bf16" if dtype == torch.bfloat16 else "fp32"
    out = {"workload": workload_text("insseg" if workload == "insseg" else workload, model_name),
           "note": note, "dtype": dname, "steps": steps, "warmup": warmup, "voxels_per_step": n_vox, "ms_per_step": ms,
           "value": n_vox * steps / res["dt"], "unit": "voxels/s", "final_loss": float(res["loss"].item()), "phases": res["phases"],
           "step_ms": dict(step_stats(res["step_ms"]), all=res["step_ms"]), "warmup_extra": res["warmup_extra"]}
    log("%s %s: %.2f ms/step; per-step %s" % (workload, dname, ms, " ".join("%.1f" % m for m in res["step_ms"])))
    if clog is not None and res["disc"] is not None:
        tp = pmc_traffic(res["disc"]["top"][0][0], args) if (workload == "clip" and dtype == torch.bfloat16) else (None, None)
        out["roofline"] = roofline_report(clog, res["disc"], dname, "clip" if workload == "clip" else ("ce" if workload == "ce" else "insseg"),
                                          steps, ms, n_vox, tp, res.get("sample_steps", 0))
        if workload == "clip":
            out["roofline"]["mfma"] = clip_mfma_report(out_dim=model.PLANES[7], n=n_vox, dtype=dtype, device=device)
    del model, ddp, opt, res
    gc.collect()
    torch.cuda.empty_cache()
    return out


def _syncbn_issuer():
    from languagegroundedsemseg_amd.ddp import EngineComm
    if not EngineComm._by_group:
        return None
    return ("the engine, on its own RCCL communicator and the compute stream (csrc/lgs_comm.hip)"
            if any(v is not None for v in EngineComm._by_group.values()) else "torch.distributed between the engine's split kernels")


def dp_path_block(coords, feats, labels, device, args, dtype, steps=8, warmup=3):
    """The code path every rank runs at N > 1 -- MinkowskiSyncBatchNorm (statistics / combine / apply as separate kernels around an
    all-gather and an all-reduce per layer) and bucketed gradient all-reduces from the backward hooks -- timed on THIS one GPU with
    RCCL and a world of one rank: the collectives are real RCCL launches the compute stream waits for, only the wire is missing.
    What a rank pays for the N > 1 path before any inter-GPU time (main.py:121-123; DESIGN section 5 budgets the wire)."""
    import gc
    if dist.is_initialized():
        return None
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    except Exception as e:          # no RCCL in this build: report, do not fail the bench line
        return {"error": "nccl process group of one rank: %s" % e}
    prev = ME.MinkowskiSyncBatchNorm.force_sync
    try:
        ME.MinkowskiSyncBatchNorm.force_sync = True
        import copy
        a2 = copy.copy(args)
        a2.dp_world1, a2.sync_bn = True, 1
        model, ddp, opt = make_trainer("Res16UNet34C", dtype, device, 1, a2, None)
        ddp.enable_timing()
        res = measure(model, ddp, opt, coords, feats, labels, dtype, steps, warmup, None, None, 1, False, base=40000)
        n_vox = int(coords.shape[0])
        out = {"note": "the per-rank code path of N > 1 (SyncBN split kernels + one all-gather and one all-reduce per layer, bucketed "
                       "gradient all-reduce) on one GPU with RCCL and a world of ONE rank: every collective is a real RCCL launch, "
                       "only the wire time is missing",
               "ms_per_step": res["dt"] / steps * 1e3, "value": n_vox * steps / res["dt"], "unit": "voxels/s", "steps": steps,
               "phases": res["phases"], "ddp": res.get("ddp"), "backend": dist.get_backend(), "allreduce": args.allreduce,
               "syncbn_collectives_issued_by": _syncbn_issuer()}
        del model, ddp, opt, res
    finally:
        ME.MinkowskiSyncBatchNorm.force_sync = prev
        from languagegroundedsemseg_amd import ddp as _ddp_mod
        _ddp_mod._TIMING["on"] = False
        _ddp_mod.EngineComm.close_all()
        dist.destroy_process_group()
    gc.collect()
    torch.cuda.empty_cache()
    return out


LINE_LIMIT = 4096      # bytes of the ONE stdout line (round 5's 21 KB line was not parsed by the driver)


def _r(x, nd=4):
    """floats rounded to `nd` significant digits for the stdout line (the details file keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def headline(full, details_path):
    """The ONE stdout line: the driver's contract keys + `roofline` + `cpu_baseline` + one number per secondary workload, under
    LINE_LIMIT bytes.  Everything else (phases, per-rank records, the discovery step, notes) is in the details file and on stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: full[k] for k in keep}
    line["value"] = _r(full["value"], 7)
    line["ms_per_step"] = _r(full["ms_per_step"], 5)
    cfg = full["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "scenes_per_gpu", "voxels_per_gpu", "global_voxels", "parallelism", "sync_bn",
                                          "allreduce", "storage") if k in cfg}
    sm = full.get("step_ms")
    if sm:
        line["step_ms"] = _r({k: sm[k] for k in ("median", "p90", "min", "max")})
        line["value_at_median"] = _r(cfg["global_voxels"] / (sm["median"] * 1e-3), 5) if full["n_gpus"] == 1 else None
    line["warmup_extra"] = full.get("warmup_extra")
    rf = full.get("roofline")
    if rf:
        line["roofline"] = _r({"bound": rf["bound"], "kernel": rf["kernel"], "achieved": rf["achieved"], "peak": rf["peak"],
                               "unit": rf["unit"], "frac": rf["frac"], "traffic": rf["traffic"],
                               "alg_bytes_per_launch": rf["alg_bytes_per_launch"], "avg_launch_ms": rf["avg_launch_ms"],
                               "launches_sampled": rf["launches_sampled"], "hbm_frac": rf["hbm_frac"],
                               "mfma_frac": rf["mfma_frac_of_peak_on_real_pairs"],
                               "step_b_alg_bytes": rf["step"]["b_alg_bytes"], "step_frac": rf["step"]["frac_of_hbm_peak"]})
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"][:200]}
    sec = {}
    for key, path in (("single_scene_ms", ("single_scene", "ms_per_step")), ("balanced_ce_ms", ("balanced", "ms_per_step")),
                      ("call_by_call_ms", ("reference_calls", "call_by_call_ms_per_step")), ("fp32_ms", ("fp32", "ms_per_step")),
                      ("fp32_step_frac", ("fp32", "roofline", "step", "frac_of_hbm_peak")),
                      ("fp32_exact_mfma_ms", ("fp32", "exact_mfma", "ms_per_step")), ("clip_ms", ("clip", "ms_per_step")),
                      ("clip_roofline_bound", ("clip", "roofline", "bound")), ("clip_roofline_frac", ("clip", "roofline", "frac")),
                      ("insseg_ms", ("insseg", "full", "ms_per_step")), ("insseg_frozen_ms", ("insseg", "frozen_trunk", "ms_per_step")),
                      ("dp_path_world1_ms", ("dp_path_world1", "ms_per_step"))):
        v = full
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
            if v is None:
                break
        if v is not None:
            sec[key] = _r(v)
    if sec:
        line["secondary"] = sec
    if full["n_gpus"] > 1:
        pr = full.get("per_rank") or []
        line["ranks"] = _r({"ms_per_step": [r["ms_per_step"] for r in pr],
                            "allreduce_exposed_wait_ms": [(r.get("ddp") or {}).get("allreduce_exposed_wait_ms") for r in pr],
                            "syncbn_collective_ms": [(r.get("ddp") or {}).get("syncbn_collective_ms") for r in pr],
                            "comm_create_s": full["rccl_ranks"].get("comm_create_s"), "backend": full["rccl_ranks"].get("backend")})
    line["details"] = details_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # never lose the record to its decorations again
        for k in ("secondary", "ranks", "step_ms", "value_at_median"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_details(full):
    """every secondary block of the run -> gpurun_out/bench_full.json (merged back by gpurun; the committed copy of the round's
    final run is profiles/rNN_bench_full.json) and, as one line, stderr.  Returns the path written (None if the tree is read-only)."""
    text = json.dumps(full)
    print("[bench details] " + text, file=sys.stderr, flush=True)
    path = os.environ.get("LGS_BENCH_DETAILS") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text + "\n")
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, script=None, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when nobody set up ranks for it (main.py:192-195:
    the reference hands `gpus=N` to Lightning, which spawns one process per GPU; here that is torch.distributed.run)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            script or os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, script=None, extra_env=None):
    """Run N ranks of `script` (this file) on this node, one per GPU, and return the launcher's exit code.  Rank 0's single
    JSON line goes to our stdout unchanged (children inherit it)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL / tensor sharing)
    env.setdefault("OMP_NUM_THREADS", "4")
    env["LGS_BENCH_SELF_LAUNCHED"] = "1"
    env.update(extra_env or {})
    return subprocess.call(launch_command(n, argv, script), env=env)


def resolve_world(gpus, environ):
    """-> ("launch", gpus) when this process must spawn the ranks itself, ("rank", world) when it IS a rank.  `--gpus N` is the
    contract: a rank whose WORLD_SIZE disagrees with it is a mis-launch and raises instead of printing n_gpus of something else."""
    if "WORLD_SIZE" not in environ:
        if gpus > 1:
            return "launch", gpus
        return "rank", 1
    world = int(environ["WORLD_SIZE"])
    if world != gpus:
        raise RuntimeError("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it as `python bench.py --gpus N` or under "
                           "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (gpus, world))
    return "rank", world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--scenes", type=int, default=8, help="synthetic scenes per GPU per step")
    ap.add_argument("--voxels", type=int, default=150000, help="target voxels per scene (@2cm)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default=None, help="default: Res16UNet34C (ce) / Res16UNet34D (clip)")
    ap.add_argument("--workload", default="ce", choices=["ce", "clip"],
                    help="ce = BASELINE configs[1] (the headline metric); clip = configs[2], CLIP-contrastive pretrain step "
                         "(scripts/text_representation_train.sh: Res16UNet34D, 512-d text anchors)")
    ap.add_argument("--settle", type=int, default=15, help="at most this many extra untimed warm-up steps (rounds of five) until two "
                                                            "rounds agree; 0 = exactly --warmup steps (profiling runs that count steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-single-scene", action="store_true", help="skip the secondary 1-scene-per-step measurement")
    ap.add_argument("--roctx", action="store_true", help="profiling runs: wrap every conv / dgrad / wgrad engine call in a ROCTx range "
                                                       "named by its layer shape (rocprofv3 --marker-trace), no HIP events")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp32 / clip (configs[2]) / insseg (configs[4]) blocks of the default line")
    ap.add_argument("--allreduce", default="ring", choices=["ring", "rs_ag"],
                    help="gradient-bucket reduction for N > 1: ring = one all-reduce per bucket; rs_ag = reduce-scatter + all-gather "
                         "on the flat bucket (all xGMI links at once, SURVEY section 5)")
    ap.add_argument("--sync-bn", type=int, default=1, help="convert to MinkowskiSyncBatchNorm when gpus > 1 (main.py:122)")
    ap.add_argument("--compute-priority", type=int, default=0, help="run the compute stream at this HIP stream priority "
                                                                     "(-1 = high: dgrad/BN chain ahead of the side-stream wgrad)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the "
                                                       "N>1 code path with several ranks on ONE GPU)")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (single-GPU dry run of the N>1 path)")
    ap.add_argument("--dp-world1", action="store_true", help="N = 1 only: run the per-rank code path of N > 1 (SyncBN collectives, bucketed "
                                                              "all-reduce) through RCCL with a world of ONE rank (diagnostic; the default "
                                                              "line carries the same measurement as `dp_path_world1`)")
    args = ap.parse_args()
    if args.model is None:
        args.model = "Res16UNet34C" if args.workload == "ce" else "Res16UNet34D"

    mode, world = resolve_world(args.gpus, os.environ)
    if mode == "launch":
        if not args.same_device and torch.cuda.device_count() < args.gpus:
            raise RuntimeError("bench.py --gpus %d: this node exposes %d GPUs" % (args.gpus, torch.cuda.device_count()))
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes its version banner to stdout when the first
    # communicator comes up.  Everything this process (and the C code it loads) writes to fd 1 goes to stderr; the line is written
    # to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (the engine has no CPU fallback)")
    if args.same_device:
        local_rank = 0
        # ranks sharing ONE GPU (dry run of the N > 1 logic only): the grid-barrier BatchNorm kernels assume that all their
        # workgroups become resident promptly, which another process filling the same CUs does not allow (measured: 697 ms per
        # step with --sync-bn 0); the three-launch path has no such assumption
        os.environ.setdefault("LGS_BN_FUSED", "0")      # initial value of the engine's BN_FUSED knob (csrc/lgs_tuning.hip)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm_create_s = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL / tensor sharing)
        t_pg = time.perf_counter()
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=device)
            except TypeError:                                       # older torch: no device_id argument
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(args.backend)
        # the first collective brings the communicator up (RCCL builds its rings / trees lazily): time it apart from the steps
        warm = torch.zeros(1, device=device)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        comm_create_s = time.perf_counter() - t_pg
        log("process group + first collective: %.2f s" % comm_create_s)
    if args.dp_world1:
        if world != 1:
            raise RuntimeError("--dp-world1 is a single-GPU diagnostic (N > 1 runs that path anyway)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
        ME.MinkowskiSyncBatchNorm.force_sync = True
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    # ---- data: scenes shard over ranks (rank r owns seeds r*S .. r*S+S-1); resident in HBM before timing
    seeds = [rank * args.scenes + i for i in range(args.scenes)]
    coords_np, feats_np, labels_np = make_batch(seeds, voxel=0.02, n_target=args.voxels)
    if os.environ.get("LGS_BENCH_SORT") == "1":   # experiment only: spatially sorted input rows (the headline keeps the dataset's arbitrary order)
        from languagegroundedsemseg_amd.synthetic import morton_order
        perm = morton_order(coords_np)
        coords_np, feats_np, labels_np = coords_np[perm], feats_np[perm], labels_np[perm]
    coords = torch.from_numpy(coords_np).to(device)
    feats = torch.from_numpy(feats_np).to(device)
    labels = torch.from_numpy(labels_np).to(device)
    n_vox = int(coords.shape[0])
    log("data resident: %d voxels in %d scenes" % (n_vox, args.scenes))

    ctx = make_ctx(args.workload, args.model, coords, device)
    model, ddp, opt = make_trainer(args.model, dtype, device, world, args, ctx)

    if args.compute_priority != 0:
        hp = torch.cuda.Stream(device=device, priority=args.compute_priority)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    import gc
    gc.collect()
    gc.disable()   # no cyclic-GC pauses inside the timed region (tensors are freed by refcount as usual)
    clog = None
    if rank == 0 and not args.no_roofline:
        clog = ConvLog()
        clog.patch()
    if args.roctx:
        import ctypes
        rl = ConvLog()
        rl.patch()
        rl.roctx = ctypes.CDLL("librocprofiler-sdk-roctx.so")
        rl.roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
        rl.mode = "roctx"
    if world > 1 or args.dp_world1:
        ddp.enable_timing()
    res = measure(model, ddp, opt, coords, feats, labels, dtype, args.steps, args.warmup, ctx, clog, world, not args.no_roofline,
                  settle=args.settle)
    dt, loss = res["dt"], res["loss"]
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    nv = torch.tensor([float(n_vox)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nv, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_vox = float(nv.item())
    ms_per_step = dt / args.steps * 1e3
    value = total_vox * args.steps / dt
    final_loss = float(loss.item())
    log("timed region done: %.2f ms/step, %.3g voxels/s" % (ms_per_step, value))
    log("per-step ms (compute-stream events): " + " ".join("%.1f" % m for m in res["step_ms"]))
    log("phases (ms per step, compute stream): " + ", ".join("%s %.2f" % kv for kv in res["phases"]["stream_ms"].items()))
    log("phases (ms per step, host enqueue time): " + ", ".join("%s %.2f" % kv for kv in res["phases"]["host_enqueue_ms"].items()))

    out = {
        "metric": "voxels/sec fwd+bwd Res16UNet34C @2cm ScanNet200", "value": value, "unit": "voxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload_text(args.workload, args.model),
                   "scenes_per_gpu": args.scenes, "voxels_per_gpu": n_vox, "global_voxels": int(total_vox),
                   "parallelism": "dp%d" % world, "sync_bn": bool(world > 1 and args.sync_bn),
                   "allreduce": args.allreduce,
                   "storage": "bf16 features / fp32 master weights, fp32 accumulate + BN statistics" if args.dtype == "bf16"
                   else "fp32",
                   # what issues the engine calls of the timed steps (round-4 review: say which model the number belongs to)
                   "model_impl": "languagegroundedsemseg_amd.models: the reference's call sequence (resnet_block.py:41-57, "
                                 "res16unet.py:196-270: conv; norm; relu in place; out += residual; me.cat) with standard "
                                 "MinkowskiEngine signatures only -- the reference's unchanged model files record the same units "
                                 "(tests/test_deferred_cpu.py); fused behind the ME surface by deferred execution (me/deferred.py), "
                                 "LGS_DEFER=%d" % int(_deferred.ENABLED),
                   "timed_steps_instrumented": False},
        "final_loss": final_loss,
        "phases": res["phases"],
        "step_ms": dict(step_stats(res["step_ms"]), all=res["step_ms"]),
        "warmup_ms": res["warmup_ms"], "warmup_extra": res["warmup_extra"],
    }
    # one record per rank (also at N = 1, so that the schema of the line does not depend on N): where a step spends its time
    # (compute-stream phases, the part of the bucket all-reduces that backward did not hide, the compute-stream stalls inside
    # SyncBN's small collectives)
    mine = {"rank": rank, "phases": res["phases"], "ddp": res.get("ddp"), "ms_per_step": res["dt"] / args.steps * 1e3,
            "step_ms": step_stats(res["step_ms"]), "warmup_extra": res["warmup_extra"]}
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        assert dist.get_world_size() == args.gpus
    else:
        allr = [mine]
    out["per_rank"] = allr
    out["rccl_ranks"] = {"world_size": dist.get_world_size() if world > 1 else 1,
                         "backend": dist.get_backend() if world > 1 else None,
                         "self_launched": os.environ.get("LGS_BENCH_SELF_LAUNCHED") == "1",
                         "bucket_collectives_per_step": mine["ddp"]["bucket_collectives_per_step"] if mine["ddp"] else None,
                         "syncbn_collectives_per_step": mine["ddp"]["syncbn_collectives_per_step"] if mine["ddp"] else None,
                         "syncbn_collectives_issued_by": _syncbn_issuer()}

    if clog is not None and res["disc"] is not None:
        out["roofline"] = roofline_report(clog, res["disc"], args.dtype, args.workload, args.steps, ms_per_step, n_vox,
                                          pmc_traffic(res["disc"]["top"][0][0], args), res.get("sample_steps", 0))
        if args.workload == "clip":
            out["roofline"]["mfma"] = clip_mfma_report(out_dim=model.PLANES[7], n=n_vox, dtype=dtype, device=device)
    log("roofline pass done")
    secondary = rank == 0 and world == 1 and not args.no_secondary and args.workload == "ce" and args.dtype == "bf16" and not args.dp_world1
    if rank == 0 and world == 1 and not args.no_single_scene and args.scenes != 1:
        out["single_scene"] = single_scene_line(model, ddp, opt, dtype, device, args, ctx)
        log("single-scene line done")
    if secondary:
        # the SAME model and batch with every ME call executed as it is made (LGS_DEFER=0): what the call sequence costs without the
        # deferred surface -- separate norm, ReLU, add and concat-copy launches, one autograd node per call
        _deferred.ENABLED = False
        try:
            ub = secondary_block("ce", "Res16UNet34C", torch.bfloat16, coords, feats, labels, device, args, None, steps=5, warmup=3,
                                 note="LGS_DEFER=0: the reference call sequence executed call by call (unfused); the headline is the same "
                                      "sequence executed through the deferred ME surface")
        finally:
            _deferred.ENABLED = True
        out["reference_calls"] = {"fused_ms_per_step": ms_per_step, "call_by_call_ms_per_step": ub["ms_per_step"],
                                  "call_by_call": ub, "deferred_stats": dict(_deferred.STATS)}
        log("call-by-call block done")
        out["balanced"] = secondary_block("ce_balanced", "Res16UNet34C", torch.bfloat16, coords, feats, labels, device, args, None, steps=8,
                                          warmup=3, note="the step scripts/train_models.sh:37 runs: CrossEntropyLoss(reduction='none') + "
                                                         "sample_categories_for_balancing with the configured ratios (-1 / -1), no host sync")
        out["balanced"]["sampled"] = secondary_block("ce_balanced_sampled", "Res16UNet34C", torch.bfloat16, coords, feats, labels, device, args,
                                                     None, steps=8, warmup=3, note="same with head / common ratios 0.5: a draw without "
                                                                                   "replacement per class on the device")
        log("balanced block done")
        # the other BASELINE configurations, each a few timed steps on the same 8-scene batch, so that their numbers sit in
        # the driver's record next to the headline instead of in builder-only files
        del model, ddp, opt
        gc.collect()
        torch.cuda.empty_cache()
        out["fp32"] = secondary_block("ce", "Res16UNet34C", torch.float32, coords, feats, labels, device, args, clog, steps=6, warmup=5,
                                      note="the parity path (fp32 storage: logits within 1e-3 of the oracle).  Convolution forward / dgrad "
                                           "and weight gradients multiply on the bf16 matrix pipe with exactly split operands (x = hi + mid + lo, six "
                                           "products, fp32 accumulate; knobs FP32_SPLIT, WGRAD_F32_LDS); 1x1 layers on the exact-fp32 MFMA")
        from languagegroundedsemseg_amd import engine as _engine
        with _engine.tuning(FP32_SPLIT=0):
            ex = secondary_block("ce", "Res16UNet34C", torch.float32, coords, feats, labels, device, args, None, steps=4, warmup=2,
                                 note="FP32_SPLIT=0: every product on v_mfma_f32_32x32x2_f32 (round 4's fp32 path)")
        out["fp32"]["exact_mfma"] = {"ms_per_step": ex["ms_per_step"], "value": ex["value"], "unit": "voxels/s", "note": ex["note"]}
        log("fp32 block done")
        out["clip"] = secondary_block("clip", "Res16UNet34D", torch.bfloat16, coords, feats, labels, device, args, clog, steps=5, warmup=3,
                                      note="BASELINE configs[2] (scripts/text_representation_train.sh): Res16UNet34D + fused CLIP text-anchor loss")
        log("clip block done")
        out["insseg"] = {
            "full": secondary_block("insseg", "InsSegRes16UNet34C", torch.bfloat16, coords, feats, labels, device, args, None, steps=5, warmup=3,
                                    note="downstream/insseg step as the reference runs it (all parameters trained, pl_Trainer.py:81)"),
            "frozen_trunk": secondary_block("insseg_frozen", "InsSegRes16UNet34C", torch.bfloat16, coords, feats, labels, device, args, None,
                                            steps=5, warmup=3, note="BASELINE configs[4]: head on frozen pretrained features (eval-mode trunk under "
                                                                    "no_grad, only offsets_pre / bntr_offset / offsets / final are trained)")}
        log("insseg block done")
        out["dp_path_world1"] = dp_path_block(coords, feats, labels, device, args, dtype)
        log("dp-path block done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model_name=args.model, voxels=args.voxels, loss="clip" if args.workload == "clip" else "ce")
        log("cpu baseline done")
        if secondary:
            out["cpu_baseline_other_configs"] = cpu_baselines_other_configs()
            log("cpu baselines of configs[0] / configs[2] done")
    out["rccl_ranks"]["comm_create_s"] = comm_create_s
    if rank == 0:
        details = write_details(out)
        line = headline(out, details)
        sys.stdout.flush()
        os.write(json_fd, (line + "\n").encode())
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        from languagegroundedsemseg_amd.ddp import EngineComm
        EngineComm.close_all()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
