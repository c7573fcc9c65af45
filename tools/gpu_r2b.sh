#!/bin/bash
# round-2 GPU session B: parity of the fused frame kernel, bench lines, stream / trend profiles
set -o pipefail
cd /root/repo
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -15 $O/gputests.log
grep -q "pytest rc=0" $O/gputests.log || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench64.log 2>&1; echo "bench64 rc=$?"; tail -c 1500 $O/bench64.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --clips 256 --no-cpu-baseline > $O/bench256.log 2>&1; echo "bench256 rc=$?"; tail -c 1500 $O/bench256.log
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --config folder --no-cpu-baseline > $O/folder1.log 2>&1; echo "folder rc=$?"; tail -c 1800 $O/folder1.log
cd /tmp; export TMPDIR=/tmp
AEGIS_STREAM_GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/st_nograph -o st -- python3 /root/repo/tools/bench_stream.py 1500 > /root/repo/$O/stream_nograph.log 2>&1; echo "stream nograph rc=$?"
cp /tmp/st_nograph/*kernel_stats.csv /root/repo/$O/stream_nograph_kernel_stats.csv 2>/dev/null || find /tmp/st_nograph -name "*stats*" -exec cp {} /root/repo/$O/ \;
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr -- python3 /root/repo/tools/bench_trend.py > /root/repo/$O/trend.log 2>&1; echo "trend rc=$?"
find /tmp/tr -name "*kernel_stats*" -exec cp {} /root/repo/$O/trend_kernel_stats.csv \;
AEGIS_DUMP_MAPS=/root/repo/$O/maps.txt timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/st_graph -o st -- python3 /root/repo/tools/bench_stream.py 1500 > /root/repo/$O/stream_graph.log 2>&1; echo "stream graph rc=$?"
find /tmp/st_graph -name "*kernel_stats*" -exec cp {} /root/repo/$O/stream_graph_kernel_stats.csv \;
tail -5 /root/repo/$O/stream_nograph.log; tail -40 /root/repo/$O/stream_graph.log | cut -c1-200
