#!/bin/bash
# tools/ab_many.sh <variant> <reps> lib1 lib2 ... : time one ab_time.py variant with many library builds, round-robin <reps> times
V=$1; R=$2; shift 2
for rep in $(seq 1 $R); do for l in "$@"; do RF_LIB=$PWD/rapidfuzz_rs_amd/$l python tools/ab_time.py $V 2>/dev/null | tail -1; done; done
