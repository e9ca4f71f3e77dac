set -u
R=$PWD; export TMPDIR=/tmp; W=/tmp/kt_dist; rm -rf $W; mkdir -p $W
cd /tmp
RF_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $W/kt.log 2>&1
cd $R
python - $W/kt/kt_results.db <<'PY'
import sqlite3,sys
cur=sqlite3.connect(sys.argv[1]).cursor()
for name,calls,total,avg,pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if calls>=20: print(f"{name[:90]:90s} {calls:5d} {avg:10.2f}us {pct:6.2f}%")
rows=list(cur.execute("select name,start,end from kernels order by start"))
# last 3 steps timeline
idx=[i for i,r in enumerate(rows) if 'stream_kernel' in r[0] or 'scan_kernel' in r[0]]
i0=idx[-3]
t0=rows[i0][1]
for r in rows[i0:idx[-1]+8]:
    print(f"{(r[1]-t0)/1e3:10.1f} {(r[2]-r[1])/1e3:9.1f}us  {r[0][:80]}")
PY
