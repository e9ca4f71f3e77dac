#!/bin/bash
# scratch: slot-store mode of the asm tiles kernels: parity (multitile forced-path modes, gather tests), A/B, randomized campaign on the gather path
set -u
mkdir -p gpurun_out/s3
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_asm_kernel.py -m gpu -x -q -n 4 -k "gather or many_tiles or mid_size or query_lengths_ragged or osa_ragged or cutoff_length_window or distinct_lengths or large_scan or asm" 2>&1 | grep -v "^  File\|^Extension" | tail -4) > gpurun_out/s3/gputests7.log 2>&1
cat gpurun_out/s3/gputests7.log
{
export AB_MINLEN=1 AB_N=100000000
bash tools/ab_many.sh lev32rag 3 librfgpu_OLD.so librfgpu.so
AB_N=20000000 bash tools/ab_many.sh lev32rag 2 librfgpu_OLD.so librfgpu.so
} > gpurun_out/s3/slot_store_asm.txt 2>&1
cat gpurun_out/s3/slot_store_asm.txt
fz() { echo "== fuzz $*"; env "$@" RF_FUZZ_SEEDS=6000 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -x -k "randomized" 2>&1 | grep -v "^\.\|^$" | tail -4; }
{ fz RF_UNSCATTER_MIN=1 RF_TILE_ORDER=0; } > gpurun_out/s3/fuzz6.log 2>&1
cat gpurun_out/s3/fuzz6.log
