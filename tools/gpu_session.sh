#!/bin/bash
# scratch: full GPU suite on the current build, then the LCS-family grid A/B (tiles per wavefront 3 under a cap of 512 against the default 5 / 256)
set -u
mkdir -p gpurun_out/s3
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/s3/gputests2.log 2>&1
cat gpurun_out/s3/gputests2.log
export AB_MINLEN=1 RF_LIB=$PWD/rapidfuzz_rs_amd/librfgpu.so
run() { echo -n "cap=$1 tiles_per_wave=$2 n=${AB_N:-default} "; RF_SCAN_BLOCKS_PER_CU_FULL=$1 RF_SCAN_TILES_PER_WAVE=$2 python tools/ab_time.py $3 2>/dev/null | tail -1; }
{
for rep in 1 2 3; do run 256 5 indel; run 512 3 indel; run 512 2 indel; run 1024 3 indel; done
for rep in 1 2; do run 256 5 lev32; run 512 3 lev32; done
export AB_N=100000000
for rep in 1 2; do run 256 5 indelrag; run 512 3 indelrag; done
} > gpurun_out/s3/lcs_grid.txt 2>&1
cat gpurun_out/s3/lcs_grid.txt
