#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q -k "band or banded or cutoff or known" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_sel.log
for v in lev256c8; do RF_NO_BAND=1 python tools/ab_time.py $v 2>/dev/null | tail -1; python tools/ab_time.py $v 2>/dev/null | tail -1; done
