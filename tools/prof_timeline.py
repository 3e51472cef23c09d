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


if __name__ == "__main__":
    main()
