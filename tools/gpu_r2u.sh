#!/bin/bash
cd /root/repo
O=gpurun_out/r2u; mkdir -p $O
timeout -k 5 60 tools/_build/ubench_walk
timeout -k 10 300 python -m pytest tests -m gpu -x -q  2>&1 | tail -2
AEGIS_BALANCED_CHUNK=0 AEGIS_TIME_CHUNK=65536 AEGIS_HIP_LIB=/root/repo/_ablate/lib_ab128.so timeout -k 10 200 python3 tools/frame_cycles.py 2>&1 | tail -8
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
run c64 AEGIS_X=0; run c64_1chunk AEGIS_BALANCED_CHUNK=0 AEGIS_TIME_CHUNK=65536
