#!/bin/bash
# scratch: ragged cutoff-3 through length-run views + multi-word asm scans (round 4)
set -u
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "query_lengths_ragged or device_rows or full_size_c3 or many_tiles_per_wavefront" -n 4 2>&1 | tail -5
echo "== C3 asm / compiled"
for env in "RF_X=1" "RF_ASM_BLOCK=0"; do env $env python bench.py --query-len 256 --cand-len 256 --candidates 10000000 --steps 10 --cpu-seconds 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'])"; done
for q in 128 192; do for env in "RF_X=1" "RF_ASM_BLOCK=0"; do echo "q$q $env"; env $env python bench.py --query-len $q --cand-len 128 --candidates 20000000 --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "length_run_views or cutoff_length_window or cutoff_early_out or test_ragged_results" 2>&1 | tail -5
for mn in 1 57; do
  for env in "RF_HEAD8_MIN=0" "RF_X=1"; do
    echo "== ragged min-len $mn cutoff 3 $env"
    env $env python bench.py --ragged --min-len $mn --cutoff 3 --steps 20 --cpu-seconds 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'])"
  done
done
echo "== single-length cutoff 3"; python bench.py --cutoff 3 --steps 20 --cpu-seconds 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'])"
