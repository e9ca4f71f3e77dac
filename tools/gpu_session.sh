#!/bin/bash
set -u
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "topk or sharded or bench_extra" 2>&1 | tail -4
