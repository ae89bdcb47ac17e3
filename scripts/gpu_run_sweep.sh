#!/bin/bash
# blocks per persistent workgroup (BFIR_PAIR_RUN): 1 = one transform per workgroup with the new kernels' other changes
set -o pipefail
OUT=gpurun_out/${1:-runs}
mkdir -p $OUT
for r in 1 2 4 8 16 32 64; do
  BFIR_PAIR_RUN=$r timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r$r.json 2>>$OUT/err.log || { echo "r$r failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/r$r.json")); r=d["roofline"]
print("run=%s value %.0f ms/set %.4f overlapped %s exclusive %s" % ("$r", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done
