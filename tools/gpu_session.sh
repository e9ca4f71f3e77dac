#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests/ -q -m gpu -n 4 2>&1 | tail -15
