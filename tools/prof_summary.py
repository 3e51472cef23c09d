"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as a per-kernel table, one row per
(kernel template instance, launch grid, workgroup): a template instance serves several layer shapes, the grid tells them
apart (k_conv_gather: grid.x = positions / tile rows, grid.y = output-channel tiles, grid.z = 3 for the slot split), so
the dominant launch shape's average duration can be read off directly.
    python tools/prof_summary.py gpurun_out/prof/x_results.db <profiled steps = warmup + timed> > profiles/rNN_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("lgs::", "")
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    return name[:86]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(end-start), avg(end-start), min(end-start), "
                       "max(end-start) from kernels group by name, grid_x, grid_y, grid_z, workgroup_x order by 7 desc").fetchall()
    tot = sum(r[6] for r in rows)
    print("# rocprofv3 --kernel-trace summary grouped by (kernel, grid, workgroup); %d groups, total %.3f ms over %.0f profiled "
          "steps (%.3f ms/step of kernel time, all streams)" % (len(rows), tot / 1e6, steps, tot / 1e6 / steps))
    print("# grid = workgroups per dimension (rocprof reports work-items; divided by the workgroup size here)")
    print("%-88s %-14s %5s %9s %10s %9s %9s %9s %6s" % ("kernel", "grid(wg)", "wg", "calls/st", "ms/step", "avg_us", "min_us", "max_us", "pct"))
    for name, gx, gy, gz, wx, n, t, a, mn, mx in rows[:90]:
        g = "%dx%dx%d" % (gx // max(wx, 1), gy, gz)
        print("%-88s %-14s %5d %9.1f %10.3f %9.1f %9.1f %9.1f %6.2f" % (short(name), g, wx, n / steps, t / 1e6 / steps, a / 1e3, mn / 1e3,
                                                                        mx / 1e3, 100.0 * t / tot))


if __name__ == "__main__":
    main()
