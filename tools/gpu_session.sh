#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
RF_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/bench_sharded1.log 2>&1; tail -1 gpurun_out/bench_sharded1.log
timeout 900 python bench.py --config c5 > gpurun_out/bench_c5.log 2>&1; tail -1 gpurun_out/bench_c5.log
