#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q -k "bench" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu_sel.log
