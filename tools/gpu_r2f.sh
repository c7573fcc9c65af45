#!/bin/bash
# CU partition experiments + kernel timeline of a step
set -o pipefail
cd /root/repo
O=gpurun_out/r2f; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"])
PY
}
run default AEGIS_X=0 && run nosplit AEGIS_CU_SPLIT=0 && run frame256 AEGIS_CU_FRAME=256 && run frame224 AEGIS_CU_FRAME=224 && run frame208 AEGIS_CU_FRAME=208 || exit 1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o tl -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/tl.log 2>&1; echo "trace rc=$?"
F=$(find /tmp/p_tl -name "*kernel_trace.csv" | head -1)
python3 - "$F" > /root/repo/$O/timeline.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "aegis::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
vit = [i for i, r in enumerate(rows) if "viterbi" in r["Kernel_Name"]]
last11 = vit[-11:]
prev = [i for i in range(last11[0]) if "decode" in rows[i]["Kernel_Name"] or "rake_runs" in rows[i]["Kernel_Name"]]
first_idx = (max(prev) + 1) if prev else 0
t0 = int(rows[first_idx]["Start_Timestamp"])
for r in rows[first_idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aegis::", "")[:28]
    print(f"{n:30s} start {(int(r['Start_Timestamp'])-t0)/1e6:9.3f} ms  end {(int(r['End_Timestamp'])-t0)/1e6:9.3f} ms  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f}")
PY
cat /root/repo/$O/timeline.txt
