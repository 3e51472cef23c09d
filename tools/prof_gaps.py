"""Where does the compute stream idle?  From a rocprofv3 --kernel-trace result (rocpd sqlite .db) take one steady-state step
(delimited by the CE kernel: loss launch = end of forward, gradient launch = start of backward) and list, for the busiest
queue, the largest gaps between consecutive kernels with the kernels on either side, plus busy / idle totals per phase.
    python tools/prof_gaps.py gpurun_out/prof/x_results.db [step_index]"""
import sqlite3
import sys

from prof_timeline import family


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
    ce = [r for r in rows if "k_ce_fwd_bwd" in r[0]]
    q = ce[0][1]
    marks = [r[2] for r in ce]
    fwd_end, next_fwd_end = marks[2 * which], marks[2 * which + 2]
    bwd_start = marks[2 * which + 1]
    mine = [r for r in rows if r[1] == q]
    # the step's forward starts with the first conv after the previous step's optimiser kernels: take the window between CE marks
    prev = [r for r in mine if marks[2 * which - 1] <= r[2] < fwd_end]
    sgd = [i for i, r in enumerate(prev) if "k_sgd_step" in r[0] or "k_pack_weights_batch" in r[0]]
    start_i = sgd[-1] + 1 if sgd else 0
    fwd = prev[start_i:]
    bwd = [r for r in mine if bwd_start <= r[2] < next_fwd_end]
    sgd2 = [i for i, r in enumerate(bwd) if "k_pack_weights_batch" in r[0]]
    bwd = bwd[:sgd2[0] + 1] if sgd2 else bwd
    for tag, seg in (("forward", fwd), ("backward+optimizer", bwd)):
        busy = sum(e - s for _, _, s, e in seg) / 1e6
        span = (seg[-1][3] - seg[0][2]) / 1e6
        gaps = [((seg[i + 1][2] - seg[i][3]) / 1e3, family(seg[i][0]), family(seg[i + 1][0]), (seg[i][3] - seg[i][2]) / 1e3) for i in range(len(seg) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        print("== %s: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms (%d gaps, median %.1f us)" % (
            tag, len(seg), span, busy, span - busy, len(pos), sorted(g[0] for g in pos)[len(pos) // 2] if pos else 0.0))
        hist = [0, 0, 0, 0, 0]
        for g in pos:
            hist[0 if g[0] < 2 else 1 if g[0] < 5 else 2 if g[0] < 10 else 3 if g[0] < 50 else 4] += 1
        print("   gaps <2 us: %d, 2-5: %d, 5-10: %d, 10-50: %d, >50: %d" % tuple(hist))
        for g in sorted(gaps, key=lambda t: -t[0])[:14]:
            print("   gap %8.1f us after %-22s (%7.1f us) before %s" % (g[0], g[1], g[3], g[2]))


if __name__ == "__main__":
    main()
