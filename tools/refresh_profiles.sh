#!/bin/bash
# Regenerates everything under profiles/ that comes from the GPU box (run through gpurun from the repo root):
#   rocprofv3 kernel-trace + PMC summaries of every BASELINE.json config's dominant kernel, and one bench.py JSON line per config.
# Results land in gpurun_out/profiles/; copy them into profiles/ afterwards.
set -u
R=${1:-r06}
mkdir -p gpurun_out/profiles
cp profiles/traffic.json gpurun_out/traffic.json 2>/dev/null
# ONLY=<substring> restricts the run to the profiles / bench lines whose tag contains it (e.g. ONLY=ragged)
# PROFILES="tag tag ..." restricts the rocprofv3 summaries (five passes each) to those tags; the bench lines always run
P() { tag=$1; key=$2; shift 2; [[ -n "${ONLY:-}" && $tag != *$ONLY* ]] && return; [[ -n "${PROFILES:-}" && " $PROFILES " != *" $tag "* ]] && return; tools/profile_c2.sh ${tag}_$R "$key" "$@"; }
MATCH="rf::stream_lev64" P c2_levenshtein "levenshtein:q64:n100000000:l64:cutNone:many"
MATCH="rf::head_filter" P c2_levenshtein_cutoff3 "levenshtein:q64:n100000000:l64:cut3:many" --cutoff 3
MATCH="rf::stream_levw4" P c3_levenshtein_256 "levenshtein:q256:n10000000:l256:cutNone:many" --query-len 256 --cand-len 256 --candidates 10000000
MATCH="rf::stream_lcs6" P c4_indel "indel:q64:n100000000:l64:cutNone:many" --metric indel
MATCH="rf::jaro" P c4_jaro_winkler "jaro_winkler:q64:n100000000:l64:cutNone:many" --metric jaro_winkler
MATCH="rf::stream_osa" P osa "osa:q64:n100000000:l64:cutNone:many" --metric osa
MATCH="rf::stream_lev32" P q32_levenshtein "levenshtein:q32:n100000000:l64:cutNone:many" --query-len 32
MATCH="rf::head_filter" P c5_cutoff3_topk "levenshtein:q64:n100000000:l64:cut3:topk" --cutoff 3 --mode topk
MATCH="rf::lev1_asm" P topk16_nocutoff "levenshtein:q64:n100000000:l64:cutNone:topk" --mode topk
MATCH="rf::band" P c3_cutoff8_band "levenshtein:q256:n10000000:l256:cut8:many" --query-len 256 --cand-len 256 --candidates 10000000 --cutoff 8
MATCH="rf::stream_lev64" P ragged_levenshtein "levenshtein:q64:n100000000:l64:cutNone:many:ragged" --ragged
MATCH="rf::window_gather" P ragged_gather "none" --ragged --metric indel
MATCH="rf::jaro" P ragged_jaro_winkler "jaro_winkler:q64:n100000000:l64:cutNone:many:ragged" --ragged --metric jaro_winkler
MATCH="rf::head_filter" P ragged_cutoff3 "levenshtein:q64:n100000000:l64:cut3:many:ragged" --ragged --min-len 57 --cutoff 3
MATCH="rf::stream_lcs6" P ragged_indel "indel:q64:n100000000:l64:cutNone:many:ragged" --ragged --metric indel
MATCH="rf::stream_levw8" P levenshtein_512 "levenshtein:q512:n2500000:l512:cutNone:many" --query-len 512 --cand-len 512 --candidates 2500000
MATCH="rf::scan_multi" P multi4_levenshtein "levenshtein:q64:n100000000:l64:cutNone:many:x4" --queries 4
# round 6: the cutoff path on a corpus that shares prefixes with the query (lane compaction), the compact (index, score) result, slot-order results
MATCH="rf::head_filter" P survivors1_cutoff3 "none" --cutoff 3 --head-share 0.01
MATCH="sparse_lean" P survivors5_cutoff3_sparse "none" --cutoff 3 --head-share 0.05
MATCH="rf::head_filter" P filter_cutoff3 "none" --cutoff 3 --mode filter
MATCH="rf::stream_lcs6" P ragged_indel_slots "none" --ragged --metric indel --slot-order
# ... long queries on corpora of near-duplicates: the band pass that lists what it leaves, the scan over that list (per-lane chunk loads: what do the counters say they cost?)
MATCH="rf::band_list" P hint16_neardup90_band "none" --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.9 --hint 16
MATCH="sparse_words" P hint16_neardup90_sparse "none" --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.9 --hint 16
MATCH="rf::band_sparse" P c3_cutoff8_neardup1_sparse "none" --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.01 --cutoff 8
sed -i "s#gpurun_out/#profiles/#g" gpurun_out/traffic.json; cp gpurun_out/traffic.json gpurun_out/profiles/traffic.json
b() { name=$1; shift; [[ -n "${ONLY:-}" && $name != *$ONLY* ]] && return; python bench.py --traffic off --extras off "$@" 2>/dev/null | tail -1 > gpurun_out/profiles/bench_$name.json; }
# the default line exactly as the driver runs it (extra_configs legs, in-run traffic)
[[ -z "${ONLY:-}" || c2_levenshtein == *$ONLY* ]] && python bench.py 2>/dev/null | tail -1 > gpurun_out/profiles/bench_c2_levenshtein.json
b ragged_cutoff3 --ragged --cutoff 3
b ragged57_cutoff3 --ragged --min-len 57 --cutoff 3
b ragged57_osa_cutoff3 --ragged --min-len 57 --cutoff 3 --metric osa --no-cpu-baseline
b q128_levenshtein --query-len 128 --cand-len 128 --candidates 20000000
b ragged_levenshtein --ragged
b ragged_q32_levenshtein --ragged --query-len 32
b ragged_osa --ragged --metric osa --no-cpu-baseline
b ragged_indel --ragged --metric indel
b ragged_jaro_winkler --ragged --metric jaro_winkler
b q32_levenshtein --query-len 32
b c3_levenshtein_256 --query-len 256 --cand-len 256 --candidates 10000000
b levenshtein_512 --query-len 512 --cand-len 512 --candidates 2500000
b levenshtein_320 --query-len 320 --cand-len 320 --candidates 4000000
b hint16_neardup99 --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.99 --hint 16 --no-cpu-baseline
b hint16_neardup90 --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.9 --hint 16
b hint16_neardup50 --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.5 --hint 16 --no-cpu-baseline
b nohint_neardup90 --query-len 256 --cand-len 256 --candidates 10000000 --near-dup-share 0.9 --no-cpu-baseline
b c4_indel --metric indel
b q32_indel --metric indel --query-len 32
b c4_lcs_seq --metric lcs_seq
b c4_jaro --metric jaro
b c4_jaro_winkler --metric jaro_winkler
b osa --metric osa
b c5_cutoff3_many --cutoff 3
b c5_cutoff3_topk --cutoff 3 --mode topk --no-cpu-baseline
b topk16_nocutoff --mode topk --no-cpu-baseline
b multi4_levenshtein --queries 4 --no-cpu-baseline
b multi4_indel --metric indel --queries 4 --no-cpu-baseline
b wf_weights_1_2_3 --weights 1,2,3 --candidates 20000000 --steps 3 --warmup 1
b indel_cutoff12 --metric indel --cutoff 12
b osa_cutoff3 --metric osa --cutoff 3
b cutoff5_many --cutoff 5
b jw_cutoff0.9 --metric jaro_winkler --fcutoff 0.9
b c3_cutoff8 --query-len 256 --cand-len 256 --candidates 10000000 --cutoff 8
b c3_cutoff8_neardup1 --query-len 256 --cand-len 256 --candidates 10000000 --cutoff 8 --near-dup-share 0.01
b c3_cutoff8_neardup50 --query-len 256 --cand-len 256 --candidates 10000000 --cutoff 8 --near-dup-share 0.5
b filter_cutoff3 --cutoff 3 --mode filter
b survivors1_cutoff3_many --cutoff 3 --head-share 0.01
b survivors1_cutoff3_topk --cutoff 3 --head-share 0.01 --mode topk --no-cpu-baseline
b survivors1_cutoff3_filter --cutoff 3 --head-share 0.01 --mode filter
b survivors5_cutoff3_many --cutoff 3 --head-share 0.05
b zipf_cutoff3_many --cutoff 3 --zipf 1.1
b ragged_indel_slots --ragged --metric indel --slot-order
b ragged_levenshtein_slots --ragged --slot-order
b ragged_jaro_winkler_slots --ragged --metric jaro_winkler --slot-order
b ragged_lognormal_indel --ragged --metric indel --lognormal-median 24 --zipf 1.1
b ragged57_filter_cutoff3 --ragged --min-len 57 --cutoff 3 --mode filter
b jw_filter0.9 --metric jaro_winkler --fcutoff 0.9 --mode filter
if [ -n "${ONLY:-}" ]; then cp gpurun_out/*_$R.txt gpurun_out/*_$R.json gpurun_out/profiles/ 2>/dev/null; ls gpurun_out/profiles | wc -l; exit 0; fi
RF_BENCH_FORCE_DIST=1 python bench.py --steps 20 --no-cpu-baseline --traffic off --extras off 2>/dev/null | tail -1 > gpurun_out/profiles/bench_sharded_path_world1.json
python bench.py --config c5 --extras off 2>/dev/null | tail -1 > gpurun_out/profiles/bench_c5_1B_world1.json
python tools/time_mixed.py > gpurun_out/profiles/mixed_tiles_$R.txt 2>/dev/null; RF_NO_MIXED_TILES=1 python tools/time_mixed.py >> gpurun_out/profiles/mixed_tiles_$R.txt 2>/dev/null
for v in lev256c8; do RF_NO_BAND=1 python tools/ab_time.py $v 2>/dev/null | tail -1 | sed 's/librfgpu.so/RF_NO_BAND=1/'; python tools/ab_time.py $v 2>/dev/null | tail -1; done > gpurun_out/profiles/band_ab_$R.txt
for v in lev64 lev64+topk lev64+topk+out indel indel+topk jw; do python tools/ab_time.py $v 2>/dev/null | tail -1; done > gpurun_out/profiles/variants_$R.txt
# kernel timeline of the sharded step (top-16 + every distance + exchange), world size 1
( cd /tmp && RF_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace -d /tmp/kt_sh_$R -o kt -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --extras off > /tmp/kt_sh_$R.log 2>&1 )
( echo "sharded step at world size 1 (RF_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5): rocprofv3 --kernel-trace, last two steps"; python tools/timeline.py /tmp/kt_sh_$R/kt_results.db 3 ) > gpurun_out/profiles/sharded_step_$R.txt
# the clock ramp after an idle phase: per-launch duration of back-to-back plain scans with the settle phase off
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_ramp_$R -o kt -- python $OLDPWD/bench.py --steps 40 --warmup 0 --settle-ms 0 --no-cpu-baseline --traffic off --extras off > /tmp/kt_ramp_$R.log 2>&1 )
python - > gpurun_out/profiles/clock_ramp_$R.txt <<PY
import sqlite3
cur = sqlite3.connect("/tmp/kt_ramp_$R/kt_results.db").cursor()
d = [(e - s) / 1e3 for n, s, e in cur.execute("select name, start, end from kernels order by start") if ("stream_kernel" in n or "lev1_asm_kernel" in n or "stream_lev64" in n) and e - s > 1_000_000]
print("python bench.py --steps 40 --warmup 0 --settle-ms 0 under rocprofv3 --kernel-trace: duration (us) of each back-to-back scan launch after the idle set-up phase")
print(" ".join(f"{x:.0f}" for x in d))
print(f"first 5 avg {sum(d[:5]) / 5:.0f} us; launches 20+ avg {sum(d[20:]) / max(1, len(d[20:])):.0f} us -> bench.py runs --settle-ms (default 200) of untimed steps before the W warm-up steps and reports config.settle_steps")
PY
cp gpurun_out/*_$R.txt gpurun_out/*_$R.json gpurun_out/profiles/ 2>/dev/null
ls gpurun_out/profiles | wc -l
