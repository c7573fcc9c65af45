#!/bin/bash
# samples sclk / power while bench.py runs (is the Viterbi slower under the frame stage because the chip clocks down?)
cd /root/repo
O=gpurun_out/r2i; mkdir -p $O
( for i in $(seq 1 400); do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' ')"; sleep 0.05; done ) > $O/smi_pipeline.log 2>&1 &
SMI=$!
timeout -k 10 300 python bench.py --steps 60 --warmup 2 --no-cpu-baseline > $O/bench_pipeline.log 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
( for i in $(seq 1 400); do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' ')"; sleep 0.05; done ) > $O/smi_single.log 2>&1 &
SMI=$!
AEGIS_TIME_CHUNK=16384 timeout -k 10 300 python bench.py --steps 60 --warmup 2 --no-cpu-baseline > $O/bench_single.log 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_ms[^}]*}" $O/bench_pipeline.log | tr "\n" " "; echo
grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_ms[^}]*}" $O/bench_single.log | tr "\n" " "; echo
echo "--- pipeline smi"; awk 'NR%12==0' $O/smi_pipeline.log | cut -c1-260 | tail -25
echo "--- single smi"; awk 'NR%12==0' $O/smi_single.log | cut -c1-260 | tail -12
