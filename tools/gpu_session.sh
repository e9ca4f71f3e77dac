#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or bench or sample or shard or parallel or random or concurrent" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_sel.log
for v in lev64 lev64+topk lev64+topk+out indel indel+topk; do RF_TOPK_ASYNC_SAMPLE=0 python tools/ab_time.py $v 2>/dev/null | tail -1 | sed 's/librfgpu.so  /in-stream    /'; python tools/ab_time.py $v 2>/dev/null | tail -1; done
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"plain\", d[\"value\"], d[\"ms_per_step\"])"; RF_BENCH_FORCE_DIST=1 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"sharded\", d[\"value\"], d[\"ms_per_step\"])"; done
