#!/bin/bash
set -u
mkdir -p gpurun_out/r04
echo "=== JW priv"; tools/pmc_lds.sh jw_priv jaro --metric jaro_winkler
echo "=== JW shared"; RF_JARO_PRIV=0 tools/pmc_lds.sh jw_shared jaro --metric jaro_winkler
echo "=== C3 asm"; tools/pmc_lds.sh c3_asm lev --query-len 256 --cand-len 256 --candidates 10000000
echo "=== C3 compiled"; RF_ASM_BLOCK=0 tools/pmc_lds.sh c3_compiled scan_kernel --query-len 256 --cand-len 256 --candidates 10000000
cp gpurun_out/pmc_*.txt gpurun_out/r04/
