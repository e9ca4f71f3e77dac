#!/bin/bash
# tools/pmc_variant.sh <ab_time variant> : SQ / GRBM counters of the scan kernel (largest grid) for one variant
set -u
export TMPDIR=/tmp
R=$PWD; V=$1; W=/tmp/pv_$$; mkdir -p $W
cd /tmp
AB_REPS=3 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d $W/a -o p -- python $R/tools/ab_time.py $V > $W/loga 2>&1
AB_REPS=3 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $W/b -o p -- python $R/tools/ab_time.py $V > $W/logb 2>&1
AB_REPS=3 rocprofv3 --pmc FETCH_SIZE -d $W/c -o p -- python $R/tools/ab_time.py $V > $W/logc 2>&1
AB_REPS=3 rocprofv3 --pmc WRITE_SIZE -d $W/d -o p -- python $R/tools/ab_time.py $V > $W/logd 2>&1
cd $R
python - <<PY
import sqlite3
print("== $V : PMC averages per dispatch, by (kernel, grid)")
for d in "abcd":
    try:
        cur = sqlite3.connect("$W/%s/p_results.db" % d).cursor()
        rows = list(cur.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection group by kernel_name, grid_size, counter_name"))
    except Exception as e:
        print("  (pass %s failed: %s)" % (d, e)); continue
    for name, grid, cname, val, n in rows:
        if "rf::" in name and "pack" not in name and "hist" not in name:
            print(f"  {name[:48]:48s} grid {grid:9d} {cname:22s} {val:18.1f}  (n={n})")
PY
rm -rf $W
