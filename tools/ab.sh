#!/bin/bash
# tools/ab.sh <variant> ... : time each ab_time.py variant alternately with every library in AB_LIBS (default: the reference
# build librfgpu_A.so and the current librfgpu.so), twice each, on the same box (boxes differ by several percent).
LIBS=${AB_LIBS:-"librfgpu_A.so librfgpu.so"}
for v in "$@"; do
  for rep in 1 2; do
    for l in $LIBS; do RF_LIB=$PWD/rapidfuzz_rs_amd/$l python tools/ab_time.py $v 2>/dev/null | tail -1; done
  done
done
