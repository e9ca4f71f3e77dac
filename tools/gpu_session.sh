#!/bin/bash
# scratch: new tests, then randomized differential tests with many seeds under several forced paths
set -u
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "q8 or scratch_is_bounded or cached_acceleration or concurrent" 2>&1 | tail -15
fz() { echo "== fuzz $*"; env "$@" RF_FUZZ_SEEDS=20000 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -x -k "randomized" 2>&1 | grep -v "^\.\|^$" | tail -30; }
fz RF_X=0
fz RF_RUN_MIN_TILES=1 RF_HEAD8_MIN=1 RF_BAND_FILTER=1
fz RF_JOINT_MAX_TILES=0 RF_TOPK_VIA_SCORES=2
