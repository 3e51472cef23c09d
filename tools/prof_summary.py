"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as a per-kernel table:
    python tools/prof_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary; %d kernels, total %.3f ms over %.0f profiled steps (%.3f ms/step)" % (
        len(rows), tot / 1e6, steps, tot / 1e6 / steps))
    print("%-100s %7s %10s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for name, n, t, a, mn, mx in rows:
        print("%-100s %7d %10.3f %10.1f %10.1f %10.1f %6.2f" % (name[:100], n, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))


if __name__ == "__main__":
    main()
