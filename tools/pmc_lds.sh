#!/bin/bash
# tools/pmc_lds.sh <tag> <kernel-match> [bench.py flags...] : LDS counters of one workload (own --pmc pass, kernel trace for the name and time)
set -u
TAG=$1; MATCH=$2; shift 2
R=$PWD; export TMPDIR=/tmp; W=/tmp/rfpmc_$TAG; rm -rf $W; mkdir -p $W gpurun_out
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $W/p1 -o p1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras off --traffic off "$@" > $W/p1.log 2>&1
rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --extras off --traffic off "$@" > $W/kt.log 2>&1
cd $R
python - "$MATCH" $W/p1/p1_results.db $W/kt/kt_results.db > gpurun_out/pmc_$TAG.txt <<'PY'
import sqlite3, sys
match = sys.argv[1]
cur = sqlite3.connect(sys.argv[2]).cursor()
for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (f"%{match}%",)):
    print(f"{c:28s} {v:18.0f}  n={n}  {k[:90]}")
cur = sqlite3.connect(sys.argv[3]).cursor()
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:100]:100s} {calls:6d} x {avg / 1e3:12.1f} us avg {pct:6.2f} %")
PY
cat gpurun_out/pmc_$TAG.txt
rm -rf $W
