#!/bin/bash
# The cutoff path against corpora that share prefixes with the query: bench.py lines for the configs[4] shape at 100 M
# (score_cutoff 3: dense u32 vector / top-16 / compact pairs) with 0 .. 20 % of the candidates carrying the query's first 8..12 symbols,
# with the lane compaction (default) and with round 5's second pass over surviving tiles (RF_LANE_COMPACT=0), same box.
#   tools/survivor_curve.sh [out-file] [candidates]
OUT=${1:-gpurun_out/survivors.txt}
N=${2:-100000000}
: > "$OUT"
line() {  # label, env, args...
    local label=$1 envs=$2; shift 2
    local json
    json=$(env $envs python bench.py --candidates "$N" --steps 20 --warmup 3 --extras off --traffic off --cpu-seconds 1 --settle-ms 100 "$@" 2>/dev/null | grep '^{' | tail -1)
    python - "$label" "$envs" "$json" "$@" >> "$OUT" <<'PY'
import json, sys
label, envs, js = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(js)
    par = d.get("parity") or {}
    print(f"{label:34s} {envs:18s} {d['value']:9.2f} Gpairs/s  {d['ms_per_step']:8.4f} ms/step  parity {par.get('mismatches','-')}/{par.get('checked','-')}"
          f"  {('pairs ' + str(d['config'].get('filter_count'))) if 'filter_count' in d['config'] else ''}   # python bench.py {' '.join(sys.argv[4:])}")
except Exception as exc:
    print(f"{label:34s} {envs:18s} FAILED {exc} {js[:200]}")
PY
}
for p in 0 0.001 0.01 0.05 0.2; do
    for lane in 1 0; do
        line "cutoff3 dense    head-share $p" "RF_LANE_COMPACT=$lane" --cutoff 3 --head-share $p
        line "cutoff3 top-16   head-share $p" "RF_LANE_COMPACT=$lane" --cutoff 3 --head-share $p --mode topk
    done
    line "cutoff3 filter   head-share $p" "RF_LANE_COMPACT=1" --cutoff 3 --head-share $p --mode filter
done
for p in 0 0.01 0.2; do
    for lane in 1 0; do
        line "ragged[57,64] cutoff3 share $p" "RF_LANE_COMPACT=$lane" --cutoff 3 --head-share $p --ragged --min-len 57
    done
    line "ragged[57,64] filter  share $p" "RF_LANE_COMPACT=1" --cutoff 3 --head-share $p --ragged --min-len 57 --mode filter
done
cat "$OUT"
