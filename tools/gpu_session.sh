#!/bin/bash
# scratch: one wavefront per workgroup for the compiled Jaro kernels on ragged corpora (RF_JARO_WPB=1) against four
set -u
mkdir -p gpurun_out/s3
export AB_MINLEN=1 RF_LIB=$PWD/rapidfuzz_rs_amd/librfgpu.so
run() { echo -n "wpb=$1 tiles_per_wave=$2 n=$AB_N "; RF_JARO_WPB=$1 RF_SCAN_TILES_PER_WAVE=$2 python tools/ab_time.py $3 2>/dev/null | tail -1; }
{
for n in 20000000 100000000; do export AB_N=$n
  for rep in 1 2; do run 4 5 jwrag; run 1 5 jwrag; done
  run 1 3 jwrag; run 1 8 jwrag; run 1 12 jwrag
done
export AB_N=20000000
run 4 5 jwragc9; run 1 5 jwragc9
} > gpurun_out/s3/jaro_wpb.txt 2>&1
cat gpurun_out/s3/jaro_wpb.txt
