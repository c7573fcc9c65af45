#!/bin/bash
# HBM counters of one bench config (default shard), separate passes -> gpurun_out/pmc_<config>/
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc_one.sh shard'
set -o pipefail
W=${1:-shard}
O=/root/repo/gpurun_out/pmc_$W; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
export AEGIS_VITERBI_PERSISTENT=0
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch_$W -o f -- python3 /root/repo/bench.py --config $W --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1; echo "pmc fetch $W rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write_$W -o w -- python3 /root/repo/bench.py --config $W --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1; echo "pmc write $W rc=$?"
F=$(find /tmp/p_fetch_$W -name "*counter_collection.csv" | head -1); Wr=$(find /tmp/p_write_$W -name "*counter_collection.csv" | head -1)
python3 /root/repo/tools/summarize_pmc.py "$F" "$Wr" 2 $O/pmc_hbm.json > $O/pmc_summary.log 2>&1; echo "pmc summary rc=$?"; cat $O/pmc_hbm.json
