#!/bin/bash
# constant-size chunk schedules
set -o pipefail
cd /root/repo
O=gpurun_out/r2h; mkdir -p $O
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
for st in 256 512 768 1024; do for rk in 0 4 64; do
  run s${st}_r${rk} AEGIS_CHUNK_START=$st AEGIS_CHUNK_GROWTH=100 AEGIS_RAMP_K=$rk AEGIS_CHUNK_TAIL=0 AEGIS_TIME_CHUNK=$st || exit 1
done; done
run s512_g110 AEGIS_CHUNK_START=512 AEGIS_CHUNK_GROWTH=110 AEGIS_CHUNK_TAIL=256
run s256_g110 AEGIS_CHUNK_START=256 AEGIS_CHUNK_GROWTH=110 AEGIS_CHUNK_TAIL=256
