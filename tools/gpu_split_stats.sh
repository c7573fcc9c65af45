#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/split_probe.py (rank $1 of $3, 5 calls) -> gpurun_out/$2 (kernel_stats csv)
R=${1:-0}; OUT=${2:-split_kernel_stats.csv}; W=${3:-8}
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_st
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_st -o st -- python3 $GRAFT_REPO_ROOT/tools/split_probe.py $R $W > $O/$OUT.log 2>&1; echo "stats rc=$?"
F=$(find /tmp/p_st -name "*kernel_stats.csv" | head -1)
cp "$F" $O/$OUT; head -25 $O/$OUT | cut -c1-160
