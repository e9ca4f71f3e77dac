#!/bin/bash
# scratch: forced-path runs of the whole GPU suite + randomized differential tests with many seeds (round-4 final validation)
set -u
mkdir -p gpurun_out/r04
run() { tag=$1; shift; echo "== $tag: $*"; env "$@" timeout 2400 python -m pytest tests/ -q -m gpu -n 4 -x -k "not one_billion and not forced and not edits_in_the_head and not length_run_views and not many_tiles and not full_size" 2>&1 | tail -2; }
run views RF_HEAD8_MIN=1 RF_BAND_FILTER=1 RF_RUN_MIN_TILES=1
run gather RF_UNSCATTER_MIN=1 RF_TILE_ORDER=3
run compiled RF_ASM_BLOCK=0 RF_ASM_STREAM=0
run viascores RF_TOPK_VIA_SCORES=2 RF_JARO_PRIV=1
echo "== randomized, 1500 seeds"
RF_FUZZ_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -x -k "randomized" 2>&1 | tail -2
