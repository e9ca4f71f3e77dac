#!/bin/bash
# scratch: q32 tests + profile refresh
set -u
mkdir -p gpurun_out/profiles; cp profiles/traffic.json gpurun_out/traffic.json
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "asm_chunk or randomized_single or c1 or query_lengths or topk" -n 4 2>&1 | tail -1
MATCH="rf::lev32_asm" tools/profile_c2.sh q32_levenshtein_r02 "levenshtein:q32:n100000000:l64:cutNone:many" --query-len 32
sed -i "s#gpurun_out/#profiles/#g" gpurun_out/traffic.json; cp gpurun_out/traffic.json gpurun_out/profiles/traffic.json
python bench.py --query-len 32 2>/dev/null | tail -1 > gpurun_out/profiles/bench_q32_levenshtein.json
cp gpurun_out/q32_levenshtein_r02.txt gpurun_out/q32_levenshtein_r02.json gpurun_out/profiles/
cut -c1-130 gpurun_out/profiles/bench_q32_levenshtein.json
