#!/bin/bash
set -o pipefail
cd /root/repo
O=gpurun_out/r2i; mkdir -p $O
run() {
  n=$1; shift
  env "$@" timeout -k 10 400 python bench.py $BARGS --no-cpu-baseline > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["launches_per_step"])
PY
}
BARGS="--steps 5 --warmup 2"
E="AEGIS_CHUNK_START=256 AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=64 AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=256"
run base $E && run k1 $E AEGIS_HIP_LIB=/root/repo/_ablate/lib_k1.so && run k2 $E AEGIS_HIP_LIB=/root/repo/_ablate/lib_k2.so
run s128 AEGIS_CHUNK_START=128 AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=64 AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=128
run s384 AEGIS_CHUNK_START=384 AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=64 AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=384
run g125r64 AEGIS_RAMP_K=64
run g125r64s256 AEGIS_RAMP_K=64 AEGIS_CHUNK_START=256
for v in base k1 k2; do
  if [ $v = base ]; then unset AEGIS_HIP_LIB; else export AEGIS_HIP_LIB=/root/repo/_ablate/lib_$v.so; fi
  echo "== $v"; timeout -k 10 300 python tools/frame_cycles.py 2>&1 | tail -4
done
