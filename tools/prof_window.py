"""List every kernel of every queue around the largest compute-stream gap of one step's forward (rocpd .db)."""
import sqlite3
import sys

from prof_timeline import family

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
ce = [r for r in rows if "k_ce_fwd_bwd" in r[0]]
q = ce[0][1]
lo, hi = ce[2 * which - 1][2], ce[2 * which][2]
mine = [r for r in rows if r[1] == q and lo <= r[2] < hi]
gaps = sorted(((mine[i + 1][2] - mine[i][3], i) for i in range(len(mine) - 1)), reverse=True)
for g, i in gaps[:2]:
    a, b = mine[i][3], mine[i + 1][2]
    print("== gap %.1f us on queue %d between %s and %s" % (g / 1e3, q, family(mine[i][0]), family(mine[i + 1][0])))
    for r in rows:
        if r[3] > a - 150e3 and r[2] < b + 60e3:
            print("   q%-2d %-26s start %+9.1f us  dur %8.1f us" % (r[1], family(r[0])[:26], (r[2] - a) / 1e3, (r[3] - r[2]) / 1e3))
