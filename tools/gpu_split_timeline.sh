#!/bin/bash
# kernel timeline of the last call of tools/split_probe.py (rank shard $1) under rocprofv3 --kernel-trace -> gpurun_out/$2
R=${1:-0}; OUT=${2:-split_timeline.txt}; W=${3:-8}
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o tl -- python3 $GRAFT_REPO_ROOT/tools/split_probe.py $R $W > $O/$OUT.log 2>&1; echo "trace rc=$?"
F=$(find /tmp/p_tl -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $O/$OUT <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "aegis::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("aegis::", "")
# the last call: everything after the last-but-one decode kernel
dec = [i for i, r in enumerate(rows) if nm(r) == "decode_kernel"]
first = dec[-2] + 1 if len(dec) >= 2 else 0
rows = rows[first:]
t0 = int(rows[0]["Start_Timestamp"])
# merge runs of the same kernel that follow each other closely
out = []
for r in rows:
    n, a, b = nm(r), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if out and out[-1][0] == n and n in ("frame_yin_kernel", "pyin_obs_kernel", "viterbi_band_kernel", "chunk_signal_kernel"):
        out[-1][2] = max(out[-1][2], b); out[-1][3] += 1; out[-1][4] += b - a
    else:
        out.append([n, a, b, 1, b - a])
for n, a, b, k, busy in out:
    print(f"{n:34s} {a:8.2f} -> {b:8.2f} ms  x{k:<3d} busy {busy:7.2f}")
PY
cat $O/$OUT.log | tail -4; cat $O/$OUT
