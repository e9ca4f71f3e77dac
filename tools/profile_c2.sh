#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py workload on the GPU box and leaves only the small summaries:
#   tools/profile_c2.sh <tag> <traffic-key> [bench.py flags...]
# kernel-trace+stats in one run; PMC counters in their own runs (never combined with tracing domains).
set -u
TAG=$1; KEY=$2; shift 2
R=$PWD; export TMPDIR=/tmp; W=/tmp/rfprof_$TAG; rm -rf $W; mkdir -p $W gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --traffic off --extras off "$@" > $W/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $W/p1 -o p1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --traffic off --extras off "$@" > $W/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $W/p2 -o p2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --traffic off --extras off "$@" > $W/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $W/p3 -o p3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --traffic off --extras off "$@" > $W/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $W/p4 -o p4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --traffic off --extras off "$@" > $W/p4.log 2>&1
cd $R
python tools/rocpd_summary.py --kernel-trace $W/kt/kt_results.db --pmc $W/p1/p1_results.db $W/p2/p2_results.db $W/p3/p3_results.db $W/p4/p4_results.db \
   --match "${MATCH:-rf::s}" --out gpurun_out/$TAG --traffic-key "$KEY" --traffic-json gpurun_out/traffic.json \
   --note "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline $* ; PMC passes with --steps 3" > gpurun_out/$TAG.stdout 2>&1
rm -rf $W
