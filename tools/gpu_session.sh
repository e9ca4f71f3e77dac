#!/bin/bash
# scratch: re-runs the two timeline extracts of tools/refresh_profiles.sh
set -u
R=r02; export TMPDIR=/tmp; mkdir -p gpurun_out/profiles
( cd /tmp && RF_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace -d /tmp/kt_sh_$R -o kt -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/kt_sh_$R.log 2>&1 )
( echo "sharded step at world size 1 (RF_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5): rocprofv3 --kernel-trace, last two steps"; python tools/timeline.py /tmp/kt_sh_$R/kt_results.db 3 ) > gpurun_out/profiles/sharded_step_$R.txt
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_ramp_$R -o kt -- python $OLDPWD/bench.py --steps 40 --warmup 0 --settle-ms 0 --no-cpu-baseline > /tmp/kt_ramp_$R.log 2>&1 )
python - > gpurun_out/profiles/clock_ramp_$R.txt <<PY
import sqlite3
cur = sqlite3.connect("/tmp/kt_ramp_$R/kt_results.db").cursor()
d = [(e - s) / 1e3 for n, s, e in cur.execute("select name, start, end from kernels order by start") if ("stream_kernel" in n or "lev1_asm_kernel" in n) and e - s > 1_000_000]
print("python bench.py --steps 40 --warmup 0 --settle-ms 0 under rocprofv3 --kernel-trace: duration (us) of each back-to-back scan launch after the idle set-up phase")
print(" ".join(f"{x:.0f}" for x in d))
print(f"first 5 avg {sum(d[:5]) / 5:.0f} us; launches 20+ avg {sum(d[20:]) / max(1, len(d[20:])):.0f} us -> bench.py runs --settle-ms (default 200) of untimed steps before the W warm-up steps and reports config.settle_steps")
PY
cat gpurun_out/profiles/clock_ramp_$R.txt | tail -2; head -14 gpurun_out/profiles/sharded_step_$R.txt
