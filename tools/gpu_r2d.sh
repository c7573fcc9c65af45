#!/bin/bash
# Viterbi variants: step time and per-wave cycle sections (no parity run: timing only)
set -o pipefail
cd /root/repo
O=gpurun_out/r2d; mkdir -p $O
for v in "$@"; do
  if [ $v = default ]; then unset AEGIS_HIP_LIB; else export AEGIS_HIP_LIB=/root/repo/_ablate/lib_$v.so; fi
  case $v in
    *ab64*) timeout -k 10 300 python tools/viterbi_cycles.py > $O/cycles_$v.txt 2>&1 || exit 1; echo "== $v"; cat $O/cycles_$v.txt;;
    *) timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench64_$v.log 2>&1 || exit 1
       python - <<PY
import json
l=[x for x in open("$O/bench64_$v.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$v", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"])
PY
    ;;
  esac
done
