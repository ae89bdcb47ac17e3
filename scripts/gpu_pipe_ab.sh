#!/bin/bash
# A/B the two-stage (BFIR_PIPE=2) and three-stage pipeline schedules at a few chunk sizes.
set -o pipefail
OUT=gpurun_out/${1:-pipe}
mkdir -p $OUT
for c in ${CHUNKS:-128 256 512}; do
  for p in 2 3; do
    BFIR_PIPE=$p timeout -k 10 300 python bench.py --chunk $c --steps 6 --warmup 2 --no-cpu-baseline > $OUT/p${p}_c${c}.json 2>> $OUT/err.log || { echo "pipe $p chunk $c failed"; tail -5 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/p${p}_c${c}.json")); r=d["roofline"]
print("pipe $p chunk $c value %.0f  ms/step %.4f  dom %s %.4f ms" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"]))
PY
  done
done
