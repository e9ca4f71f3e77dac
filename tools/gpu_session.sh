#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_sh -o kt -- python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline > /tmp/kt_sh.log 2>&1 )
python - <<'PY'
import sqlite3
cur = sqlite3.connect("/tmp/kt_sh/kt_results.db").cursor()
rows = list(cur.execute("select name, start, end, grid_x, stream_id, queue_id from kernels order by start"))
scans = [(s, e) for n, s, e, g, st, q in rows if "stream_kernel" in n and e - s > 1_000_000]
for i in range(len(scans) - 1):
    print(i, f"dur {(scans[i][1]-scans[i][0])/1e3:8.1f}  period {(scans[i+1][0]-scans[i][0])/1e3:8.1f}")
PY
