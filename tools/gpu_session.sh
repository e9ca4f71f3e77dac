#!/bin/bash
# scratch: Jaro partial chunks as 4-column groups: parity tests, then A/B against the old rf_jaro.hip (librfgpu_JA.so)
set -u
mkdir -p gpurun_out/s3
{
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py tests/test_asm_kernel.py -q -m gpu -x -k "jaro or Jaro or jw or winkler or randomized or f64" -n 4 2>&1 | tail -5
export AB_MINLEN=1 AB_LIBS="librfgpu_JA.so librfgpu.so"
AB_N=100000000 bash tools/ab.sh jwrag
AB_N=20000000 bash tools/ab.sh jwrag jarorag jwragc9
bash tools/ab.sh jw
} > gpurun_out/s3/jaro_cols.txt 2>&1
cat gpurun_out/s3/jaro_cols.txt
