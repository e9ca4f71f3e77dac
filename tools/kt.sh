#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass over a bench.py command line, top kernels to stdout:  tools/kt.sh [rows] -- <bench.py flags>
ROWS=${1:-10}; shift; shift
R=$PWD; export TMPDIR=/tmp; W=/tmp/kt_$$; rm -rf $W; mkdir -p $W
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $W -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --traffic off --extras off "$@" > $W/log 2>&1 )
grep '^{' $W/log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench:', d['value'], 'Gpairs/s', d['ms_per_step'], 'ms/step')"
python $R/tools/top_kernels.py $W/kt_results.db $ROWS | grep -v "at::native"
rm -rf $W
