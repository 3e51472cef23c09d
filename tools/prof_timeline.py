"""Per-queue view of a rocprofv3 --kernel-trace result (rocpd sqlite .db): for the steady-state steps (delimited by
the CE kernel) print each HIP queue's busy time per step and its kernels by family.
    python tools/prof_timeline.py gpurun_out/prof/x_results.db [skip_steps]"""
import collections
import re
import sqlite3
import sys


def family(name):
    m = re.search(r"lgs::(k_\w+)", name)
    if m:
        return m.group(1)
    if "rocprim" in name:
        return "rocprim"
    if "at::native" in name:
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
        f = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|CatArray\w+|reduce_kernel)", name)
        return "torch:" + (f.group(1) if f else (m.group(1) if m else "?"))
    return name[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
    marks = [r[2] for r in rows if "k_ce_fwd_bwd" in r[0]]
    # two CE launches per step (loss in forward, gradient in backward): use every second one
    marks = marks[::2]
    if len(marks) < skip + 2:
        print("not enough steps"); return
    t0, t1 = marks[skip], marks[-1]
    nsteps = len(marks) - 1 - skip
    sel = [r for r in rows if t0 <= r[2] < t1]
    print("# %d steady steps, %.3f ms/step wall (under the profiler)" % (nsteps, (t1 - t0) / 1e6 / nsteps))
    perq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for name, q, s, e in sel:
        d = perq[q][family(name)]
        d[0] += 1; d[1] += (e - s) / 1e6
    for q, fam in sorted(perq.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in fam.values())
        print("\nqueue %s: busy %.3f ms/step, %d launches/step" % (q, tot / nsteps, sum(v[0] for v in fam.values()) / nsteps))
        for f, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
            print("   %-44s %6.1f launches/step %8.3f ms/step  avg %7.1f us" % (f, n / nsteps, t / nsteps, 1e3 * t / n))
    # where the busiest queue (the compute stream) is IDLE: gaps between consecutive kernels, by (kernel before -> kernel after)
    main_q = max(perq.items(), key=lambda kv: sum(v[1] for v in kv[1].values()))[0]
    mine = [r for r in sel if r[1] == main_q]
    gaps = collections.defaultdict(lambda: [0, 0.0])
    idle = 0.0
    for a, b in zip(mine, mine[1:]):
        g = (b[2] - a[3]) / 1e6
        if g <= 0:
            continue
        idle += g
        d = gaps[(family(a[0]), family(b[0]), "long" if g > 0.05 else "short")]
        d[0] += 1; d[1] += g
    print("\nqueue %s idle: %.3f ms/step between its kernels; by (before -> after), gaps > 50 us apart from the rest" % (main_q, idle / nsteps))
    for (fa, fb, kind), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:16]:
        print("   %-24s -> %-24s %-5s %6.1f gaps/step %8.3f ms/step  avg %7.1f us" % (fa, fb, kind, n / nsteps, t / nsteps, 1e3 * t / n))
    # the step's head: from the end of the previous step's last compute kernel to the first convolution of the step,
    # and what the other queues ran meanwhile (the coordinate manager's insert + first maps)
    heads = []
    for i in range(skip, len(marks) - 1):
        step = [r for r in mine if marks[i] <= r[2] < marks[i + 1]]
        # the step boundary on the compute queue: the optimizer's kernels follow the second CE launch; the next step's first conv follows them
        convs = [j for j, r in enumerate(step) if "k_conv_gather" in r[0]]
        sgd = [j for j, r in enumerate(step) if "k_sgd_step" in r[0] or "k_pack_weights" in r[0]]
        if not convs or not sgd:
            continue
        last_opt = max(sgd)
        nxt = [j for j in convs if j > last_opt]
        if nxt:
            heads.append((step[nxt[0]][2] - step[last_opt][3]) / 1e6)
    # busy time of the compute queue per PHASE of the step (compare with bench.py's `phases.stream_ms`, measured without the
    # profiler: phase time - busy time = idle inside that phase): forward = first conv after the optimizer .. the loss launch,
    # backward = the loss-gradient launch .. the last kernel before the optimizer
    all_ce = [r[2] for r in mine if "k_ce_fwd_bwd" in r[0]]
    ph = collections.defaultdict(float)
    nph = 0
    for i in range(skip, len(marks) - 1):
        step = [r for r in mine if marks[i] <= r[2] < marks[i + 1]]
        ces = [j for j, r in enumerate(step) if "k_ce_fwd_bwd" in r[0]]
        sgd = [j for j, r in enumerate(step) if "k_sgd_step" in r[0]]
        if len(ces) < 2 or not sgd:
            continue
        nph += 1
        for j, r in enumerate(step):
            d = (r[3] - r[2]) / 1e6
            if j < ces[1]:
                ph["(loss launch, counted with backward of the previous mark)"] += 0
            if ces[1] <= j < min(sgd):
                ph["backward (loss gradient .. before the optimizer)"] += d
            elif min(sgd) <= j <= max(sgd) + 1:
                ph["optimizer (+ weight re-pack)"] += d
            elif j > max(sgd) + 1:
                ph["forward of the NEXT step (input side .. loss)"] += d
            elif j < ces[1]:
                ph["forward tail (loss launch)"] += d
    if nph:
        print("\nbusy time of queue %s by phase:" % main_q)
        for k, v in ph.items():
            if v:
                print("   %-60s %8.3f ms/step" % (k, v / nph))
    if heads:
        print("\nstep head (end of the optimizer's last kernel -> first convolution of the next step): avg %.3f ms, min %.3f, max %.3f" % (
            sum(heads) / len(heads), min(heads), max(heads)))


if __name__ == "__main__":
    main()
