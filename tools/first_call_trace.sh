#!/bin/bash
# which kernels of a bench.py run are SLOW ONCE (the first call's structures): the ten longest single dispatches.   tools/first_call_trace.sh <bench.py flags>
R=$PWD; export TMPDIR=/tmp; W=/tmp/fct_$$; rm -rf $W; mkdir -p $W
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $W -o kt -- python $R/bench.py --steps 3 --warmup 0 --settle-ms 0 --no-cpu-baseline --traffic off --extras off "$@" > $W/log 2>&1 )
python - $W/kt_results.db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
print("ten longest dispatches (ms, start ms, name):")
for n, s, e in sorted(rows, key=lambda r: r[1] - r[2])[:10]:
    print(f"{(e - s) / 1e6:10.3f} {(s - t0) / 1e6:10.1f}  {n[:110]}")
rf = [(n, s, e) for n, s, e in rows if "rf::" in n]
print("the library's dispatches in order (first 40): duration ms, gap to the previous one ms")
prev = rf[0][1]
for n, s, e in rf[:40]:
    print(f"{(e - s) / 1e6:10.3f} {(s - prev) / 1e6:10.3f}  {n[:100]}")
    prev = e
PY
rm -rf $W
