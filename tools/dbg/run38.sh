cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d $O/profg -o x -- python $R/bench.py --no-cpu-baseline --no-roofline --no-single-scene --no-secondary --steps 20 --warmup 6 > $O/profg.log 2>&1
DB=$(find $O/profg -name "*.db" | head -1)
cd $R/tools && python prof_window.py $DB 12 > $O/window.txt 2>&1
python - "$DB" >> $O/window.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print([r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'").fetchall() if 'copy' in r[0].lower() or 'memory' in r[0].lower()])
PY
rm -rf $O/profg
cat $O/window.txt | head -90
