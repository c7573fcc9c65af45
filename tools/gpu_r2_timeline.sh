#!/bin/bash
cd /root/repo; O=/root/repo/gpurun_out/r2z; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_trend.py tests/test_gpu_v2_engine.py -m gpu -x -q 2>&1 | tail -2
python3 tools/bench_trend.py 2>&1 | grep "^{" | cut -c1-700
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o tl -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/tl.log 2>&1; echo "trace rc=$?"
F=$(find /tmp/p_tl -name "*kernel_trace.csv" | head -1); echo $F; wc -l $F
python3 - "$F" > $O/timeline.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "aegis::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = last 11 viterbi launches and everything after the previous step's last kernel
vit = [i for i, r in enumerate(rows) if "viterbi" in r["Kernel_Name"]]
last11 = vit[-11:]
first_idx = max(i for i in range(last11[0]) if "decode" in rows[i]["Kernel_Name"] or "rake_runs" in rows[i]["Kernel_Name"]) + 1 if any("decode" in r["Kernel_Name"] for r in rows[:last11[0]]) else 0
t0 = int(rows[first_idx]["Start_Timestamp"])
for r in rows[first_idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aegis::", "")[:28]
    print(f"{n:30s} start {(int(r['Start_Timestamp'])-t0)/1e6:9.3f} ms  end {(int(r['End_Timestamp'])-t0)/1e6:9.3f} ms  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f}  grid {r.get('Grid_Size','')}")
PY
cat $O/timeline.txt
