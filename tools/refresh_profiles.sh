#!/bin/bash
# Regenerates everything under profiles/ that comes from the GPU box (run through gpurun from the repo root):
#   rocprofv3 kernel-trace + PMC summaries of the three dominant workloads, and one bench.py JSON line per config.
set -u
R=${1:-r01}
tools/profile_c2.sh c2_levenshtein_$R "levenshtein:q64:n100000000:l64:cutNone:many"
tools/profile_c2.sh c2_levenshtein_cutoff3_$R "levenshtein:q64:n100000000:l64:cut3:many" --cutoff 3
tools/profile_c2.sh c4_indel_$R "indel:q64:n100000000:l64:cutNone:many" --metric indel
sed -i "s#gpurun_out/#profiles/#g" gpurun_out/traffic.json; mkdir -p gpurun_out/profiles && cp gpurun_out/traffic.json profiles/traffic.json
b() { name=$1; shift; python bench.py "$@" 2>/dev/null | tail -1 > gpurun_out/profiles/bench_$name.json; }
b c2_levenshtein
b q32_levenshtein --query-len 32
b c3_levenshtein_256 --query-len 256 --cand-len 256 --candidates 10000000
b c4_indel --metric indel
b c4_lcs_seq --metric lcs_seq
b c4_jaro --metric jaro
b c4_jaro_winkler --metric jaro_winkler
b osa --metric osa
b c5_cutoff3_many --cutoff 3
b c5_cutoff3_topk --cutoff 3 --mode topk --no-cpu-baseline
b multi4_levenshtein --queries 4 --no-cpu-baseline
b multi4_indel --metric indel --queries 4 --no-cpu-baseline
cp gpurun_out/c2_* gpurun_out/c4_* gpurun_out/traffic.json gpurun_out/profiles/ 2>/dev/null
ls -la gpurun_out/profiles
b wf_weights_1_2_3 --weights 1,2,3 --candidates 20000000 --steps 3 --warmup 1
b indel_cutoff12 --metric indel --cutoff 12
b osa_cutoff3 --metric osa --cutoff 3
b jw_cutoff0.9 --metric jaro_winkler --fcutoff 0.9
RF_BENCH_FORCE_DIST=1 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/profiles/bench_sharded_path_world1.json
ls gpurun_out/profiles | wc -l
