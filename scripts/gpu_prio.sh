#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-prio}
mkdir -p $OUT
i=0
for pr in "0 0 0" "0 -1 0" "-1 0 0" "0 0 -1" "-1 0 -1" "0 -1 -1"; do
  i=$((i+1))
  BFIR_STREAM_PRIO="$pr" timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-exclusive-pass > $OUT/p$i.json 2>>$OUT/err.log || { echo "failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/p$i.json")); r=d["roofline"]
print("prio(front mac back)=%s value %.0f ms/set %.4f overlapped %s" % ("$pr", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}))
PY
done
