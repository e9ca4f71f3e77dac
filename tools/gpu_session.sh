#!/bin/bash
set -u
tools/ab_many.sh lev32 2 librfgpu.so librfgpu_a32_0x80.so librfgpu_a32_0x0.so librfgpu_a32_0x100.so librfgpu_a32_0x120.so librfgpu_a32_0x1A0.so
