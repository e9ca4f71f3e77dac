#!/bin/bash
# scratch: the tests behind the one that stopped the last full run; then single-length mid-size corpora on a persistent-size grid
set -u
mkdir -p gpurun_out/s3
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_c5 or ranks_script or real_ranks or ragged_cutoff_scans or loaded_corpus or five_streams or mid_size" 2>&1 | grep -v "^  File\|^Extension" | tail -40) > gpurun_out/s3/gputests4.log 2>&1
cat gpurun_out/s3/gputests4.log
