#!/bin/bash
# SQ counters of the analyze kernels (two passes; counter collection serialises kernels, so one Viterbi launch per chunk)
cd /tmp; export TMPDIR=/tmp AEGIS_VITERBI_PERSISTENT=0
O=/root/repo/gpurun_out/r2sq; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sq1 -o a -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/p1.log 2>&1; echo "pass1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/sq2 -o b -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/p2.log 2>&1; echo "pass2 rc=$?"
A=$(find /tmp/sq1 -name "*counter_collection.csv" | head -1); B=$(find /tmp/sq2 -name "*counter_collection.csv" | head -1)
python3 /root/repo/tools/summarize_sq.py 2 $O/sq_counters.json "$A" "$B"
