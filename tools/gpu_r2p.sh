#!/bin/bash
cd /root/repo
O=gpurun_out/r2p; mkdir -p $O
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
for b in 256 512 768 1024; do run b$b AEGIS_BALANCED_CHUNK=$b; done
run u_s256 AEGIS_BALANCED_CHUNK=0 AEGIS_CHUNK_START=256
run u_s256_g150 AEGIS_BALANCED_CHUNK=0 AEGIS_CHUNK_START=256 AEGIS_CHUNK_GROWTH=150
run u_s128_g150 AEGIS_BALANCED_CHUNK=0 AEGIS_CHUNK_START=128 AEGIS_CHUNK_GROWTH=150
run u_r64 AEGIS_BALANCED_CHUNK=0 AEGIS_RAMP_K=64
run u_s256_r64_k1024 AEGIS_BALANCED_CHUNK=0 AEGIS_CHUNK_START=256 AEGIS_RAMP_K=64 AEGIS_TIME_CHUNK=1024
