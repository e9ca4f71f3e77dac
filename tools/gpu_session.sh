#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): parity tests, the default bench line, the issue-rate
# probe and a kernel trace of the sharded step.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
timeout 600 python bench.py --query-len 256 --cand-len 256 --candidates 10000000 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; tail -1 gpurun_out/bench_c3.log
RF_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_sharded1.log 2>&1; tail -1 gpurun_out/bench_sharded1.log
cd /tmp
RF_BENCH_FORCE_DIST=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_sh -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/kt_sh.log 2>&1
cd $R
python tools/rocpd_summary.py --kernel-trace /tmp/kt_sh/kt_results.db --match "rf::s" --out gpurun_out/sharded_step_trace > /dev/null 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("/tmp/kt_sh/kt_results.db").cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
out = open("gpurun_out/sharded_step_timeline.txt", "w")
for name, s, e in rows[-40:]:
    out.write(f"{(s - t0) / 1e3:12.1f} us  +{(e - s) / 1e3:9.1f} us  {name[:100]}\n")
PY
tail -45 gpurun_out/sharded_step_timeline.txt
