#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or bench or sample or shard or parallel or random" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_sel.log
bash tools/ab.sh lev64 lev64+topk lev64+topk+out indel+topk > gpurun_out/ab.log 2>&1; cat gpurun_out/ab.log
bash tools/trace_variant.sh lev64+topk | tail -4
