#!/bin/bash
# tools/build_jaro_variant.sh NAME "<extra hipcc flags>" : rapidfuzz_rs_amd/librfgpu_NAME.so = the current objects (make first) with
# rf_jaro.hip recompiled under extra flags (-DRF_JARO_WAVES=7 -DRF_JARO_PREFETCH=0 ...).  tools/ab.sh times the libraries in AB_LIBS.
set -e  # JSRC=<path>: another rf_jaro.hip (e.g. git show HEAD:... copied INTO csrc/ so that its includes resolve)
NAME=$1; FLAGS=${2:-}
SRC=rapidfuzz_rs_amd/csrc; OBJ=/tmp/rf_jvariant_$NAME; mkdir -p $OBJ
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $FLAGS -c ${JSRC:-$SRC/rf_jaro.hip} -o $OBJ/rf_jaro.o
OTHERS=$(ls $SRC/*.o | grep -v rf_jaro.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o rapidfuzz_rs_amd/librfgpu_$NAME.so $OBJ/rf_jaro.o $OTHERS -ldl
ls -la rapidfuzz_rs_amd/librfgpu_$NAME.so
