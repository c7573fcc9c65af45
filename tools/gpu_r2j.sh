#!/bin/bash
set -o pipefail
cd /root/repo
O=gpurun_out/r2j; mkdir -p $O
run() {
  n=$1; shift
  env "$@" timeout -k 10 400 python bench.py $BARGS --no-cpu-baseline > $O/bench_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/bench_$n.log; return 1; }
  python - <<PY
import json
l=[x for x in open("$O/bench_$n.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$n", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["launches_per_step"])
PY
}
E256="AEGIS_CHUNK_START=256 AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=64 AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=256"
E512="AEGIS_CHUNK_START=512 AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=64 AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=512"
E1024="AEGIS_CHUNK_START=1024 AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=64 AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=1024"
BARGS="--steps 3 --warmup 1 --clips 256"
run c256_def AEGIS_X=0; run c256_256 $E256; run c256_512 $E512; run c256_1024 $E1024
BARGS="--steps 2 --warmup 1 --config folder"
run folder_def AEGIS_X=0; run folder_256 $E256; run folder_512 $E512; run folder_1024 $E1024
BARGS="--steps 5 --warmup 2 --clips 8"
run c8_def AEGIS_X=0; run c8_256 $E256; run c8_512 $E512
BARGS="--steps 5 --warmup 2 --clips 32"
run c32_def AEGIS_X=0; run c32_256 $E256
BARGS="--steps 3 --warmup 1 --clips 128"
run c128_def AEGIS_X=0; run c128_256 $E256; run c128_512 $E512
