#!/bin/bash
# parity suite, then step time and per-wave cycle sections of the current build
set -o pipefail
cd /root/repo
O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -8 $O/gputests.log
grep -q "pytest rc=0" $O/gputests.log || exit 1
bash tools/gpu_r2d.sh default ab64
