#!/bin/bash
# Long queries against corpora with near-duplicates of the query: bench.py lines for the configs[2] corpus shape (query 256 x 10 M x 256)
#   * under score_cutoff 8 (the small-band kernel) with 0 .. 99 % of the rows the query after 0..8 substitutions: the hand-over of sparse tiles to a dense second
#     pass (default) against the plain kernel (RF_BAND_DEFER=0);
#   * under score_hint 16: the band pass that lists what it leaves + the scan over the list (default), round 5's mark / sums / host / copy road with the credited
#     sample (RF_HINT_LISTS=0), and round 5's road as it was (RF_HINT_LISTS=0 RF_HINT_TRUST=0); no hint for comparison.
# Same box, in-run parity against the oracle on every line.   tools/neardup_curve.sh [out-file] [candidates]
OUT=${1:-gpurun_out/neardup.txt}
N=${2:-10000000}
: > "$OUT"
line() {  # label, env, args...
    local label=$1 envs=$2; shift 2
    local json
    json=$(env $envs python bench.py --query-len 256 --cand-len 256 --candidates "$N" --steps 32 --warmup 4 --extras off --traffic off --cpu-seconds 1 --settle-ms 100 "$@" 2>/dev/null | grep '^{' | tail -1)
    python - "$label" "$envs" "$json" "$@" >> "$OUT" <<'PY'
import json, sys
label, envs, js = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(js)
    par = d.get("parity") or {}
    print(f"{label:30s} {envs:34s} {d['value']:8.3f} Gpairs/s  {d['ms_per_step']:8.4f} ms/step  parity {par.get('mismatches','-')}/{par.get('checked','-')}   # python bench.py --query-len 256 --cand-len 256 {' '.join(sys.argv[4:])}")
except Exception as exc:
    print(f"{label:30s} {envs:34s} FAILED {exc} {js[:200]}")
PY
}
for p in 0 0.001 0.01 0.1 0.3 0.5 0.6 0.7 0.9 0.99; do
    for d in 1 0; do
        line "cutoff 8   near-dup $p" "RF_BAND_DEFER=$d" --cutoff 8 --near-dup-share $p
    done
done
for p in 0.5 0.7 0.8 0.9 0.99; do
    line "hint 16    near-dup $p" "RF_HINT_LISTS=1" --hint 16 --near-dup-share $p
    line "hint 16    near-dup $p" "RF_HINT_LISTS=0" --hint 16 --near-dup-share $p
    line "hint 16    near-dup $p" "RF_HINT_LISTS=0 RF_HINT_TRUST=0" --hint 16 --near-dup-share $p
    line "no hint    near-dup $p" "RF_X=0" --near-dup-share $p
done
cat "$OUT"
