#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "corpus_file or corrupt or stream" 2>&1 | tail -3
RF_STREAM_TIMING=1 python tools/time_stream.py 100000000 256 2>&1 | grep -v amdgpu.ids | tail -8
RF_STREAM_TIMING=1 python tools/time_stream.py 1000000000 256 512 2>&1 | grep -v amdgpu.ids | tail -14
