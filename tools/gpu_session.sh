#!/bin/bash
# scratch: the full GPU suite on the round's last build, then the ragged query-32 bench line
set -u
mkdir -p gpurun_out/s3 gpurun_out/profiles
(time timeout 560 python -m pytest tests -m gpu -q -n 3 2>&1 | grep -v "^  File\|^Extension" | tail -6) > gpurun_out/s3/gputests_last.log 2>&1
cat gpurun_out/s3/gputests_last.log
