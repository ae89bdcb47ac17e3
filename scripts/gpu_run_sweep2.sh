#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-runs2}
mkdir -p $OUT
for combo in "4 4" "2 2" "4 2" "2 4" "8 4" "4 8" "8 8" "1 1" "3 3" "4 4"; do
  set -- $combo
  BFIR_PAIR_RUN_FWD=$1 BFIR_PAIR_RUN_INV=$2 timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r$1_$2.json 2>>$OUT/err.log || { echo "failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/r$1_$2.json")); r=d["roofline"]
print("fwd=%s inv=%s value %.0f ms/set %.4f overlapped %s exclusive %s" % ("$1", "$2", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done
