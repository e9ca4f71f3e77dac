#!/bin/bash
# tools/ab_env.sh "<ENV=val>" <variant> ... : time each ab_time.py variant alternately without and with the environment setting,
# twice each, on the same box (boxes differ by several percent: only same-session pairs are comparable)
SET=$1; shift
for v in "$@"; do
  for rep in 1 2; do
    python tools/ab_time.py $v 2>/dev/null | tail -1 | sed "s/librfgpu.so   /default       /"
    env $SET python tools/ab_time.py $v 2>/dev/null | tail -1 | sed "s/librfgpu.so   /$SET/"
  done
done
