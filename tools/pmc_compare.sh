#!/bin/bash
# tools/pmc_compare.sh <tag> <kernel-match> [bench.py flags...] : two SQ counter passes for one workload, printed as a table
set -u
TAG=$1; MATCH=$2; shift 2
R=$PWD; export TMPDIR=/tmp; W=/tmp/rfpmc_$TAG; rm -rf $W; mkdir -p $W gpurun_out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU -d $W/p1 -o p1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $W/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH -d $W/p2 -o p2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $W/p2.log 2>&1
cd $R
python - "$MATCH" $W/p1/p1_results.db $W/p2/p2_results.db > gpurun_out/pmc_$TAG.txt <<'PY'
import sqlite3, sys
match = sys.argv[1]
for f in sys.argv[2:]:
    try:
        cur = sqlite3.connect(f).cursor()
        for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (f"%{match}%",)):
            print(f"{c:28s} {v:18.0f}  n={n}  {k[:70]}")
    except Exception as e:
        print("ERR", f, e)
PY
tail -2 $W/p1.log >> gpurun_out/pmc_$TAG.txt
rm -rf $W
