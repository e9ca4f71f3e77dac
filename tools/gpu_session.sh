#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or bench or multi or random or full_size" > gpurun_out/pytest_gpu_sel.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_sel.log
for v in lev64 indel lev64+topk+out lev256 jw lev64c3 lev256c8; do for b in 32 256; do RF_SCAN_BLOCKS_PER_CU_FULL=$b python tools/ab_time.py $v 2>/dev/null | tail -1 | sed "s/librfgpu.so  /full=$b  /"; done; done
for q in 4; do for b in 32 256; do RF_SCAN_BLOCKS_PER_CU_FULL=$b python bench.py --queries 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('multi4 full=$b', d['value'])"; done; done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench default', d['value'], d['roofline']['frac'], d['roofline']['issue_bound'])"
