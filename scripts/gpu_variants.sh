#!/bin/bash
# MAC kernel variant sweep (BFIR_MAC_VARIANT) at a few chunk sizes; prints value + k_mac avg launch ms.
set -o pipefail
TAG=${1:-var}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in ${VARIANTS:-0 1 2 3 4}; do
  for c in ${CHUNKS:-128 256}; do
    BFIR_MAC_VARIANT=$v timeout -k 10 300 python bench.py --chunk $c --steps 6 --warmup 2 --no-cpu-baseline > $OUT/v${v}_c${c}.json 2>> $OUT/err.log || { echo "variant $v chunk $c failed"; tail -5 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/v${v}_c${c}.json")); r=d["roofline"]
print("variant $v chunk $c value %.0f  k_mac ms/launch %s  shares %s" % (d["value"], r["avg_launch_ms"] if r["kernel"]=="k_mac" else "(dom=%s)"%r["kernel"], r["kernel_ms_share"]))
PY
  done
done
