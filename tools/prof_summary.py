"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as a per-kernel table, one row per
(kernel template instance, launch grid, workgroup[, layer shape]): a template instance serves several layer shapes; the
grid tells most of them apart (k_conv_gather: grid.x = positions / tile rows, grid.y = output-channel tiles, grid.z = 3 for the
slot split), but e.g. the 3^3 96->96 and 128->96 convolutions of level 0 share instance AND grid.  When the run was made with
`bench.py --roctx` under `rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace`, every conv / dgrad / wgrad engine
call is wrapped in a ROCTx range named by its shape ("lgs conv_fwd K=27 96->96 rows=1205389"); a kernel dispatch is matched
to the range that encloses its hipLaunchKernel call (kernels.stack_id -> the API region -> the enclosing marker range on that
thread), and the shape becomes part of the group key -- roofline.frac of the dominant shape is then recomputable from this
table alone (algorithmic bytes of the shape / avg_us).
    python tools/prof_summary.py gpurun_out/prof/x_results.db <profiled steps = warmup + timed> > profiles/rNN_kernel_stats.txt"""
import bisect
import json
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
    # ---- ROCTx shape ranges (optional): per thread, sorted by start
    marks = {}
    try:
        # a ROCTx range is a region of category MARKER_CORE_RANGE_API whose message sits in `extdata` ({"message": "lgs ..."})
        for name, tid, st, en, ext in cur.execute("select name, tid, start, end, extdata from regions where category like '%MARKER%'"):
            msg = name
            if ext and "message" in ext:
                try:
                    msg = json.loads(ext).get("message", name)
                except ValueError:
                    pass
            if msg.startswith("lgs "):
                marks.setdefault(tid, []).append((st, en, msg[4:]))
    except sqlite3.Error:
        marks = {}
    for tid in marks:
        marks[tid].sort()
    starts = {tid: [m[0] for m in v] for tid, v in marks.items()}
    api = {}
    if marks:
        for sid, tid, st in cur.execute("select stack_id, tid, start from regions where category like 'HIP_RUNTIME_API%' and name like 'hip%Launch%'"):
            api[sid] = (tid, st)

    def shape_of(stack_id):
        a = api.get(stack_id)
        if a is None:
            return ""
        tid, t = a
        lst = marks.get(tid)
        if not lst:
            return ""
        i = bisect.bisect_right(starts[tid], t) - 1
        while i >= 0 and lst[i][1] < t:        # ranges do not nest here; step back over ranges that ended before t
            i -= 1
            if i < 0 or t - lst[i][0] > 50_000_000:
                return ""
        return lst[i][2] if i >= 0 and lst[i][0] <= t <= lst[i][1] else ""

    groups = {}
    for name, gx, gy, gz, wx, st, en, sid in cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, start, end, stack_id from kernels"):
        key = (name, gx, gy, gz, wx, shape_of(sid) if marks else "")
        g = groups.setdefault(key, [0, 0, 1 << 62, 0])
        d = en - st
        g[0] += 1; g[1] += d; g[2] = min(g[2], d); g[3] = max(g[3], d)
    rows = sorted(((k[0], k[1], k[2], k[3], k[4], g[0], g[1], g[1] / g[0], g[2], g[3], k[5]) for k, g in groups.items()), key=lambda r: -r[6])
    tot = sum(r[6] for r in rows)
    print("# rocprofv3 --kernel-trace summary grouped by (kernel, grid, workgroup); %d groups, total %.3f ms over %.0f profiled "
          "steps (%.3f ms/step of kernel time, all streams)" % (len(rows), tot / 1e6, steps, tot / 1e6 / steps))
    print("# grid = workgroups per dimension (rocprof reports work-items; divided by the workgroup size here)")
    if marks:
        print("# shape = ROCTx range of the engine call that launched the kernel (bench.py --roctx): conv_fwd / conv_dgrad / wgrad K cin->cout rows")
    print("%-70s %-14s %5s %9s %10s %9s %9s %9s %6s  %s" % ("kernel", "grid(wg)", "wg", "calls/st", "ms/step", "avg_us", "min_us", "max_us", "pct", "shape"))
    for name, gx, gy, gz, wx, n, t, a, mn, mx, shp in rows[:140]:
        g = "%dx%dx%d" % (gx // max(wx, 1), gy, gz)
        print("%-70s %-14s %5d %9.1f %10.3f %9.1f %9.1f %9.1f %6.2f  %s" % (short(name)[:70], g, wx, n / steps, t / 1e6 / steps, a / 1e3, mn / 1e3,
                                                                            mx / 1e3, 100.0 * t / tot, shp))


if __name__ == "__main__":
    main()
