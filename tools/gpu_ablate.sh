#!/bin/bash
# Viterbi timing experiments with ablation builds: bash tools/gpu_ablate.sh <tag> <lib suffixes...>
# each build runs the 64 x 180 s workload as ONE time chunk (every kernel alone on the chip) and prints kernel_ms
set -o pipefail
cd /root/repo
O=gpurun_out/$1; shift; mkdir -p $O
for a in "$@"; do
  lib=/root/repo/_ablate/lib_ab$a.so; [ "$a" = "prod" ] && lib=/root/repo/spectrogram-midi_amd/libaegis_hip.so
  AEGIS_BALANCED_CHUNK=0 AEGIS_TIME_CHUNK=65536 AEGIS_HIP_LIB=$lib timeout -k 10 200 python bench.py --config shard --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_ab$a.log 2>&1 || { echo "ab$a failed"; tail -3 $O/bench_ab$a.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_ab$a.log") if x.startswith("{")][-1]; d=json.loads(l)
print("ab$a", d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
done
