#!/bin/bash
# tools/ab.sh <variant> ... : time each variant alternately with rapidfuzz_rs_amd/librfgpu_A.so (the reference build) and the
# current librfgpu.so, twice each, on the same box.
for v in "$@"; do
  for rep in 1 2; do
    RF_LIB=$PWD/rapidfuzz_rs_amd/librfgpu_A.so python tools/ab_time.py $v 2>/dev/null | tail -1
    python tools/ab_time.py $v 2>/dev/null | tail -1
  done
done
