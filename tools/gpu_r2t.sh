#!/bin/bash
cd /root/repo
O=gpurun_out/r2t; mkdir -p $O
run() {
  n=$1; shift
  env "$@" timeout -k 10 200 python bench.py $BARGS --no-cpu-baseline > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"])
PY
}
BARGS="--steps 8 --warmup 2"
for b in 256 320 384 448 512 640; do run b$b AEGIS_BALANCED_CHUNK=$b; done
