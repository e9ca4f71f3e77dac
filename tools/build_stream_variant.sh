#!/bin/bash
# tools/build_stream_variant.sh NAME "GEN_ENV=val ..." : builds rapidfuzz_rs_amd/librfgpu_NAME.so whose asm stream kernels come from
# tools/gen_stream_asm.py run with the given environment (experiment knobs RF_GEN_*); every other object is the current build's.
# Variant builds define RF_EXPERIMENTS: the measurement switch RF_EXP_NOHBM (wrong results on purpose) exists in them and ONLY in them.
# tools/ab_many.sh <variant> <reps> librfgpu.so librfgpu_NAME.so ... times them round-robin on one box.
set -e
NAME=$1; GENV=${2:-}
SRC=rapidfuzz_rs_amd/csrc; OBJ=/tmp/rf_svariant_$NAME; mkdir -p $OBJ
env $GENV python tools/gen_stream_asm.py $OBJ/rf_stream_asm.inc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-inline-asm -DRF_EXPERIMENTS "-DRF_STREAM_ASM_INC=\"$OBJ/rf_stream_asm.inc\"" -c $SRC/rf_stream_asm.hip -o $OBJ/rf_stream_asm.o
OTHERS=$(ls $SRC/*.o | grep -v rf_stream_asm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o rapidfuzz_rs_amd/librfgpu_$NAME.so $OTHERS $OBJ/rf_stream_asm.o -ldl
ls -la rapidfuzz_rs_amd/librfgpu_$NAME.so
