cd /root/repo
export AB_N=100000000 AB_MINLEN=1
for v in levrag+norm lev32rag+norm osarag+norm indelrag+norm; do python tools/ab_time.py $v 2>&1 | grep Gpairs; RF_NORM_GATHER=0 python tools/ab_time.py $v 2>&1 | grep Gpairs | sed 's/librfgpu.so  /norm_gather=0/'; done
unset AB_N AB_MINLEN
RF_UNSCATTER_MIN=1 RF_FUZZ_SEEDS=600 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -n 4 -q -k randomized 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "ragged or gather or unscatter or bucket or normalized" 2>&1 | tail -2
