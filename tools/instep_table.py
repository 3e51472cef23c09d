"""In-step vs stand-alone duration of every launch shape of the training step (round-4 review, item 3).

Two rocprofv3 kernel traces of the SAME workload, both summarised by tools/prof_summary.py (shape-keyed through bench.py --roctx):
  A  the production step: compute stream + weight-gradient side stream + map stream overlap;
  B  the same step with every weight gradient on the compute stream (host knob LGS_DBG_WGRAD=inline) and the maps of a step
     built before its forward starts: kernels run one after the other, i.e. each at its STAND-ALONE rate on the real tensors.
For every (kernel, grid, workgroup, shape) row:  in-step us / stand-alone us, and what the inflation costs per step
(calls x (in-step - stand-alone)).  Rows above 1.5 x on the compute stream are what a schedule change can win back.

    python tools/instep_table.py profiles/rNN_kernel_stats.txt profiles/rNN_kernel_stats_standalone.txt > profiles/rNN_instep_table.txt"""
import re
import sys


def load(path):
    rows = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel "):
            continue
        m = re.match(r"^(.{70}) (\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*(.*)$", line.rstrip("\n"))
        if not m:
            continue
        name, grid, wg, calls, ms, avg, mn, mx, pct, shape = m.groups()
        rows[(name.strip(), grid, wg, shape.strip())] = (float(calls), float(ms), float(avg))
    return rows


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    out = []
    for k, (calls, ms, avg) in a.items():
        if k in b and b[k][2] > 0:
            sa = b[k][2]
            out.append((calls * (avg - sa) / 1e3, avg / sa, avg, sa, calls, k))
    out.sort(reverse=True)
    tot_a, tot_b = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
    matched = sum(a[k][1] for _, _, _, _, _, k in out)
    print("# in-step (A: %s) vs stand-alone (B: %s); kernel time per step A %.2f ms, B %.2f ms; %.1f %% of A's kernel time matched by key" % (
        sys.argv[1], sys.argv[2], tot_a, tot_b, 100.0 * matched / max(tot_a, 1e-9)))
    print("# extra = calls per step x (in-step - stand-alone): what the co-residency costs this shape per step (negative: it runs faster in the step)")
    print("%-58s %-12s %8s %10s %10s %7s %9s  %s" % ("kernel", "grid(wg)", "calls/st", "in-step us", "alone us", "ratio", "extra ms", "shape"))
    for extra, ratio, avg, sa, calls, (name, grid, wg, shape) in out[:70]:
        print("%-58s %-12s %8.1f %10.1f %10.1f %7.2f %9.3f  %s" % (name[:58], grid, calls, avg, sa, ratio, extra, shape))
    worst = [r for r in out if r[1] > 1.5 and r[0] > 0.05]
    print("# rows above 1.5 x that cost more than 0.05 ms per step: %d, together %.2f ms per step" % (len(worst), sum(r[0] for r in worst)))


if __name__ == "__main__":
    main()
