#!/bin/bash
OUT=gpurun_out/${1:-fp64b}; mkdir -p $OUT
for blocks in 8192 32768; do for d in 1 0; do
  BFIR_DIRECT=$d timeout -k 10 300 python bench.py --workload plugin_2ch_65536tap_L1024_fp64 --blocks $blocks --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/p_${blocks}_$d.json 2>>$OUT/err.log || { echo failed; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/p_${blocks}_$d.json")); r=d["roofline"]
print("blocks=%s direct=%s value %.0f launch ms %s exclusive %s" % ("$blocks", "$d", d["value"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done; done
for d in 1 0; do echo "BFIR_DIRECT=$d"; BFIR_DIRECT=$d timeout -k 10 300 python scripts/plugin_shape.py 2>&1 | grep realsize; done
