#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
RF_BENCH_FORCE_DIST=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_sh -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/kt_sh.log 2>&1
cd $R
python - <<'PY'
import sqlite3
cur = sqlite3.connect("/tmp/kt_sh/kt_results.db").cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute("select name, start, end, grid_x, stream_id, queue_id from kernels order by start"))
t0 = rows[0][1]
prev = None
for name, s, e, g, st, q in rows[-36:]:
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{(s - t0) / 1e3:12.1f} us  +{(e - s) / 1e3:9.1f} us  gap {gap:7.1f}  grid {g:8d} stream {st} queue {q}  {name[:60]}")
    prev = e
PY
