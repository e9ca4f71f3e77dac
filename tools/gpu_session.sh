#!/bin/bash
# scratch: band tests + refresh of the band profile / bench line
set -u
mkdir -p gpurun_out/profiles; cp profiles/traffic.json gpurun_out/traffic.json
timeout 900 python -m pytest tests -q -m gpu -x -k "band or banded" 2>&1 | tail -1
MATCH="rf::band" tools/profile_c2.sh c3_cutoff8_band_r02 "levenshtein:q256:n10000000:l256:cut8:many" --query-len 256 --cand-len 256 --candidates 10000000 --cutoff 8
sed -i "s#gpurun_out/#profiles/#g" gpurun_out/traffic.json; cp gpurun_out/traffic.json gpurun_out/profiles/traffic.json
python bench.py --query-len 256 --cand-len 256 --candidates 10000000 --cutoff 8 2>/dev/null | tail -1 > gpurun_out/profiles/bench_c3_cutoff8.json
for v in lev256c8; do RF_NO_BAND=1 python tools/ab_time.py $v 2>/dev/null | tail -1 | sed 's/librfgpu.so/RF_NO_BAND=1/'; python tools/ab_time.py $v 2>/dev/null | tail -1; done > gpurun_out/profiles/band_ab_r02.txt
cp gpurun_out/c3_cutoff8_band_r02.txt gpurun_out/c3_cutoff8_band_r02.json gpurun_out/profiles/
cat gpurun_out/profiles/band_ab_r02.txt; cut -c1-130 gpurun_out/profiles/bench_c3_cutoff8.json
