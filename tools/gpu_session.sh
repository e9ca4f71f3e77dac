#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
RF_FUZZ_SEEDS=3000 timeout 2700 python -m pytest tests -m gpu -x -q -k "randomized" -n 4 > gpurun_out/pytest_fuzz.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_fuzz.log
