#!/bin/bash
# streaming MAC range: default (1024 blocks at the headline shape) vs 512 / 768, alternating on one box
OUT=gpurun_out/${1:-macrange4}; mkdir -p $OUT
for r in 0 512 768 0 512 768 0 512; do
  if [ $r = 0 ]; then unset BFIR_MAC_RANGE; else export BFIR_MAC_RANGE=$r; fi
  timeout -k 10 300 python bench.py --blocks 65536 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-exclusive-pass > $OUT/r$r.json 2>>$OUT/err.log || { echo failed; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/r$r.json")); r=d["roofline"]
print("mac range=%s value %.0f ms/set %.4f overlapped %s" % ("$r", d["value"], r["pipeline"]["ms_per_launch_set"], {k:round(v["avg_launch_ms"],3) for k,v in r["kernels"].items()}))
PY
done
