#!/bin/bash
# round-2 GPU session C: Viterbi arg-max variants -- parity, then step time per variant
set -o pipefail
cd /root/repo
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log; tail -8 $O/gputests.log
grep -q "pytest rc=0" $O/gputests.log || exit 1
for v in default vA vold; do
  if [ $v = default ]; then unset AEGIS_HIP_LIB; else export AEGIS_HIP_LIB=/root/repo/_ablate/lib_$v.so; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench64_$v.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("$O/bench64_$v.log") if x.startswith("{")][-1]; d=json.loads(l)
print("$v", d["ms_per_step"], d["value"], {k:v for k,v in d.get("kernels_ms",{}).items()} if "kernels_ms" in d else d.get("roofline"))
PY
done
unset AEGIS_HIP_LIB
AEGIS_HIP_LIB=/root/repo/_ablate/lib_ab64.so timeout -k 10 300 python tools/viterbi_cycles.py > $O/vit_cycles.txt 2>&1; cat $O/vit_cycles.txt
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --clips 256 --no-cpu-baseline > $O/bench256.log 2>&1; tail -c 600 $O/bench256.log
