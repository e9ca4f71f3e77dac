#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
AB_LIBS="librfgpu_A.so librfgpu.so librfgpu_JU.so" bash tools/ab.sh jw jaro
timeout 1500 python -m pytest tests -m gpu -x -q -k "jaro or known or random" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_sel.log
