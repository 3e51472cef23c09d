#!/bin/bash
# one-off: what does rocprofv3's rocpd database hold for kernels / marker ranges / HIP API calls (used to key the kernel
# table by layer shape in tools/prof_summary.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/schema
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format rocpd -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --roctx --steps 2 --warmup 1 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python - "$DB" > $O/schema.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
for name, sql in cur.execute("select name, sql from sqlite_master where type in ('table','view')").fetchall():
    print("====", name); print(sql)
    try:
        rows = cur.execute("select * from %s limit 3" % name).fetchall()
        cols = [d[0] for d in cur.description]
        print(cols)
        for r in rows: print(r)
        print("count", cur.execute("select count(*) from %s" % name).fetchone())
    except Exception as e:
        print("ERR", e)
PY
rm -rf $O/prof
tail -5 $O/prof.log
