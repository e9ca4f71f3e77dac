#!/bin/bash
# scratch: s_nop placement of the multi-word kernels with the carry-flag HP shift (single-bit flips of 0x0826)
set -u
mkdir -p gpurun_out/s3
{
bash tools/ab_many.sh lev256 2 librfgpu.so librfgpu_VW.so $(cd rapidfuzz_rs_amd && ls librfgpu_a0x*.so)
} > gpurun_out/s3/addc_masks.txt 2>&1
cat gpurun_out/s3/addc_masks.txt
