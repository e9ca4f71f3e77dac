#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>" : builds rapidfuzz_rs_amd/librfgpu_NAME.so from the working tree with extra
# compile flags (compile-time A/B experiments; tools/ab.sh picks the libraries named in AB_LIBS).
set -e
NAME=$1; FLAGS=${2:-}
SRC=rapidfuzz_rs_amd/csrc; OBJ=/tmp/rf_variant_$NAME; mkdir -p $OBJ
for f in rf_api rf_scan rf_long rf_jaro rf_pack rf_probe rf_select rf_mixed rf_band rf_lev_asm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o rapidfuzz_rs_amd/librfgpu_$NAME.so $OBJ/*.o -ldl
ls -la rapidfuzz_rs_amd/librfgpu_$NAME.so
