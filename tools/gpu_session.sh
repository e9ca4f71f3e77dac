#!/bin/bash
# scratch: parity of the build with ScanParams::slot_store: the gather-path tests, then the randomized campaigns with every corpus on the gather path
set -u
mkdir -p gpurun_out/s3
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "gather or many_tiles or mid_size or query_lengths_ragged or osa_ragged or cutoff_length_window or distinct_lengths or large_scan or full_size_c3" 2>&1 | grep -v "^  File\|^Extension" | tail -4) > gpurun_out/s3/gputests6.log 2>&1
cat gpurun_out/s3/gputests6.log
fz() { echo "== fuzz $*"; env "$@" RF_FUZZ_SEEDS=6000 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -x -k "randomized" 2>&1 | grep -v "^\.\|^$" | tail -4; }
{ fz RF_UNSCATTER_MIN=1; fz RF_UNSCATTER_MIN=1 RF_TILE_ORDER=0 RF_GATHER_WINDOWS=0; } > gpurun_out/s3/fuzz5.log 2>&1
cat gpurun_out/s3/fuzz5.log
