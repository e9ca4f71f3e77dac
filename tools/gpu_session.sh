#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or allgather or selfcheck" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu_sel.log
