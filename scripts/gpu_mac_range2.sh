#!/bin/bash
OUT=gpurun_out/${1:-macrange}; mkdir -p $OUT
for r in 0 256 512 2048 4096; do
  if [ $r = 0 ]; then unset BFIR_MAC_RANGE; else export BFIR_MAC_RANGE=$r; fi
  timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r$r.json 2>>$OUT/err.log || { echo failed; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/r$r.json")); r=d["roofline"]
print("mac range=%s value %.0f ms/set %.4f exclusive %s" % ("$r", d["value"], r["pipeline"]["ms_per_launch_set"], r.get("exclusive_launch_ms")))
PY
done
