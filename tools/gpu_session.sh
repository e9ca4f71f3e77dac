#!/bin/bash
set -u
mkdir -p gpurun_out/r04
echo "== W=4 nop masks round 2 (lev256, 10 M x 256)"
tools/ab_many.sh lev256 2 librfgpu_w0x0926.so librfgpu_w0x0927.so librfgpu_w0x0924.so librfgpu_w0x0922.so librfgpu_w0x092E.so librfgpu_w0x0936.so librfgpu_w0x0906.so librfgpu_w0x0966.so librfgpu_w0x09A6.so librfgpu_w0x0826.so librfgpu_w0x0B26.so librfgpu_w0x0D26.so librfgpu_w0x0126.so librfgpu_w0x1926.so | sort -k4 -n
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "length_run_views" 2>&1 | tail -5
