#!/bin/bash
set -u
mkdir -p gpurun_out/r04
tools/ab_many.sh lev32 2 librfgpu.so librfgpu_o32_1.so librfgpu_o32_2.so librfgpu_o32_3.so librfgpu_o32_4.so librfgpu_o32_5.so
