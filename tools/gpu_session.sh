#!/bin/bash
# scratch: full GPU suite on the current build, then randomized differential campaigns under forced paths
set -u
mkdir -p gpurun_out/s3
(time timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/s3/gputests3.log 2>&1
cat gpurun_out/s3/gputests3.log
fz() { echo "== fuzz $*"; env "$@" RF_FUZZ_SEEDS=6000 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -x -k "randomized" 2>&1 | grep -v "^\.\|^$" | tail -6; }
{ fz RF_X=0; fz RF_RUN_MIN_TILES=1 RF_HEAD8_MIN=1 RF_BAND_FILTER=1; fz RF_SCAN_TILES_PER_WAVE=2 RF_SCAN_BLOCKS_PER_CU_FULL=1; } > gpurun_out/s3/fuzz3.log 2>&1
cat gpurun_out/s3/fuzz3.log
