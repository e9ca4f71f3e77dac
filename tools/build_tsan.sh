#!/bin/bash
# Builds a ThreadSanitizer variant of the library (host code instrumented; the device code is what it always is) and the C++ stress
# driver tests/cpp/stress_threads.cpp against it, under build/tsan/.  Run on a GPU box:
#   tools/build_tsan.sh && TSAN_OPTIONS="suppressions=tools/tsan.supp history_size=4" build/tsan/stress_threads
# The HIP runtime is not instrumented: reports whose stacks lie entirely inside libamdhip64 / libhsa-runtime64 are suppressed.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/build/tsan
mkdir -p $O
cd $R/rapidfuzz_rs_amd/csrc
SRCS=$(grep '^SRCS' Makefile | sed 's/^SRCS *[:+]*= *//')
for s in $SRCS; do
  o=$O/${s%.hip}.o
  if [ ! -f $o ] || [ $s -nt $o ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=thread -Wno-unused-function -c $s -o $o & fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=thread -o $O/librfgpu.so $O/*.o -ldl
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=thread -I $R/include $R/tests/cpp/stress_threads.cpp -o $O/stress_threads -L $O -lrfgpu -Wl,-rpath,$O -Wl,-rpath,/opt/rocm/lib -lpthread
echo built $O/stress_threads
