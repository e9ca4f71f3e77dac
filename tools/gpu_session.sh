#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q -k "distinct_lengths or file or stream or u32 or relabel" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
