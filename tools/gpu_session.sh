#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_LIBS="librfgpu.so librfgpu_GW.so" bash tools/ab.sh lev64 indel osa lev32 > gpurun_out/ab.log 2>&1; cat gpurun_out/ab.log
timeout 900 python -m pytest tests -m gpu -x -q -k "allgather or selfcheck" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_sel.log
