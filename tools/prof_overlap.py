"""When do the map-stream kernels of step t+1 run relative to the compute stream's step t?  (rocpd sqlite .db of a
rocprofv3 --kernel-trace run of bench.py)
    python tools/prof_overlap.py x_results.db"""
import sqlite3
import sys

from prof_timeline import family

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
packs = [r for r in rows if "k_pack_weights_batch" in r[0]]          # last kernel of a step on the compute stream
keys = [r for r in rows if "k_pack_keys" in r[0]]                     # first kernel of an insert (map stream)
maps3 = [r for r in rows if "k_build_map3" in r[0]]
ce = [r for r in rows if "k_ce_fwd_bwd" in r[0]][::2]
print("compute queue", packs[0][1], " map queue", keys[0][1])
for i in range(8, min(len(packs) - 1, 14)):
    end_prev = packs[i][3]
    nxt_keys = [k for k in keys if k[2] > packs[i - 1][3]][0]
    first_conv = [r for r in rows if r[1] == packs[0][1] and r[2] > end_prev and ("k_conv_gather" in r[0] or "k_pad_rows" in r[0])][0]
    m3 = [k for k in maps3 if k[2] > packs[i - 1][3]][:5]
    print("step %d: insert starts %+8.1f us, L0 3^3 map build ends %+8.1f us, first conv starts %+8.1f us  (relative to the end of the previous step on the compute stream)" % (
        i, (nxt_keys[2] - end_prev) / 1e3, (m3[0][3] - end_prev) / 1e3 if m3 else float('nan'), (first_conv[2] - end_prev) / 1e3))
