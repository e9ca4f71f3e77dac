#!/bin/bash
# tools/build_scan_variants.sh "<flag-prefix>" v1 v2 ... : librfgpu_<v>.so for each value, recompiling ONLY rf_scan.hip with
# <flag-prefix><v> (e.g. -DRF_NOPMASK=) and linking the other objects of the normal build.  8 compilations in parallel.
set -e
PFX=$1; shift
SRC=rapidfuzz_rs_amd/csrc
make -C $SRC -j8 >/dev/null
OTHERS=$(ls $SRC/*.o | grep -v rf_scan.o)
build_one() {
  v=$1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $PFX$v -c $SRC/rf_scan.hip -o /tmp/rf_scan_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o rapidfuzz_rs_amd/librfgpu_v$v.so /tmp/rf_scan_$v.o $OTHERS -ldl
}
n=0
for v in "$@"; do build_one $v & n=$((n+1)); if [ $((n % 8)) = 0 ]; then wait; fi; done
wait
ls rapidfuzz_rs_amd/librfgpu_v*.so | wc -l
