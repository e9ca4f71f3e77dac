#!/bin/bash
# tools/kernel_names.sh <cmd...>: which kernels a command launches (rocprofv3 kernel trace), with call counts and total time.
# Run from the repo root; relative paths under tools/ and bench.py are made absolute (rocprofv3 runs from /tmp).
export TMPDIR=/tmp; W=/tmp/kn_$$; R=$PWD
args=(); for a in "$@"; do case "$a" in tools/*|bench.py|tests/*) args+=("$R/$a");; *) args+=("$a");; esac; done
cd /tmp
rocprofv3 --kernel-trace --stats -d $W -o kn -- "${args[@]}" > $W.log 2>&1
python - <<PY
import sqlite3
cur = sqlite3.connect("$W/kn_results.db").cursor()
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:100]:100s} {calls:6d} x {avg:12.1f} us avg {pct:6.2f} %")
PY
rm -rf $W $W.log
