#!/bin/bash
# tools/trace_variant.sh <ab_time variant> : rocprofv3 kernel trace of tools/ab_time.py <variant>, prints the last dispatches
set -u
export TMPDIR=/tmp
R=$PWD; V=$1; W=/tmp/tv_$$; mkdir -p $W
cd /tmp
rocprofv3 --kernel-trace --stats -d $W -o kt -- python $R/tools/ab_time.py $V > $W/log 2>&1
cd $R
python - <<PY
import sqlite3
cur = sqlite3.connect("$W/kt_results.db").cursor()
rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
t0 = rows[0][1]
print("== $V : last 12 dispatches (start us, duration us, gap to previous end us, grid, name)")
prev = None
for name, s, e, g in rows[-12:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:12.1f} {(e - s) / 1e3:9.1f} {gap:7.1f} {g:8d}  {name[:70]}")
    prev = e
PY
rm -rf $W
