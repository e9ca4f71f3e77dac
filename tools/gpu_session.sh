#!/bin/bash
# scratch: Jaro per-tile costs under the new grid policy: prefetch (JP), global common/len2 table (JT), div3 (JV), none (JB), all (librfgpu.so)
set -u
mkdir -p gpurun_out/s3
{
export AB_MINLEN=1 AB_LIBS="librfgpu_JB.so librfgpu.so librfgpu_JP.so librfgpu_JT.so librfgpu_JV.so"
AB_N=20000000 bash tools/ab.sh jwrag jarorag
AB_N=100000000 bash tools/ab.sh jwrag
bash tools/ab.sh jw
} > gpurun_out/s3/jaro_tile.txt 2>&1
cat gpurun_out/s3/jaro_tile.txt
