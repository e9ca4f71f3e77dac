#!/bin/bash
# scratch: refresh the q32 profile + bench line only
set -u
mkdir -p gpurun_out/profiles; cp profiles/traffic.json gpurun_out/traffic.json
tools/profile_c2.sh q32_levenshtein_r02 "levenshtein:q32:n100000000:l64:cutNone:many" --query-len 32
sed -i "s#gpurun_out/#profiles/#g" gpurun_out/traffic.json; cp gpurun_out/traffic.json gpurun_out/profiles/traffic.json
python bench.py --query-len 32 2>/dev/null | tail -1 > gpurun_out/profiles/bench_q32_levenshtein.json
cp gpurun_out/q32_levenshtein_r02.txt gpurun_out/q32_levenshtein_r02.json gpurun_out/profiles/
cut -c1-300 gpurun_out/profiles/bench_q32_levenshtein.json
