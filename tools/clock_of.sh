#!/bin/bash
# tools/clock_of.sh <kernel name substring> <cmd...>: the core clock a kernel actually ran at = GRBM_GUI_ACTIVE per XCD / its duration
# (two rocprofv3 passes: --pmc GRBM_GUI_ACTIVE, and --kernel-trace; run from the repo root)
export TMPDIR=/tmp; W=/tmp/clk_$$; R=$PWD; M=$1; shift
args=(); for a in "$@"; do case "$a" in tools/*|bench.py|tests/*) args+=("$R/$a");; *) args+=("$a");; esac; done
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $W/p -o p -- "${args[@]}" > $W.log 2>&1
rocprofv3 --kernel-trace --stats -d $W/k -o k -- "${args[@]}" >> $W.log 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$W/p/p_results.db").cursor()
v = [r[0] for r in c.execute("select value from counters_collection where kernel_name like ? and counter_name='GRBM_GUI_ACTIVE'", ("%$M%",))]
k = sqlite3.connect("$W/k/k_results.db").cursor()
d = [r[0] for r in k.execute("select duration from kernels where name like ? order by start", ("%$M%",))]
v, d = v[len(v)//2:], d[len(d)//2:]
cyc, ns = sum(v) / len(v) / 8, sum(d) / len(d)
print(f"$M: {cyc/1e6:.3f} M cycles per XCD, {ns/1e3:.1f} us -> {cyc/ns:.3f} GHz  ({len(v)} / {len(d)} dispatches)")
PY
rm -rf $W $W.log
