#!/bin/bash
# scratch: refresh the OSA profile + bench lines only
set -u
mkdir -p gpurun_out/profiles; cp profiles/traffic.json gpurun_out/traffic.json
tools/profile_c2.sh osa_r02 "osa:q64:n100000000:l64:cutNone:many" --metric osa
sed -i "s#gpurun_out/#profiles/#g" gpurun_out/traffic.json; cp gpurun_out/traffic.json gpurun_out/profiles/traffic.json
python bench.py --metric osa 2>/dev/null | tail -1 > gpurun_out/profiles/bench_osa.json
python bench.py --metric osa --cutoff 3 2>/dev/null | tail -1 > gpurun_out/profiles/bench_osa_cutoff3.json
cp gpurun_out/osa_r02.txt gpurun_out/osa_r02.json gpurun_out/profiles/
cut -c1-200 gpurun_out/profiles/bench_osa.json
