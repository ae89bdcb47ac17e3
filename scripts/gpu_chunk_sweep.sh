#!/bin/bash
# blocks-per-launch sweep of the default configuration
set -o pipefail
OUT=gpurun_out/${1:-chunks}
mkdir -p $OUT
for c in ${CHUNKS:-64 128 256 512 1024}; do
  timeout -k 10 300 python bench.py --chunk $c --steps ${STEPS:-6} --warmup 2 --blocks ${BLOCKS:-1024} --no-cpu-baseline > $OUT/c${c}.json 2>> $OUT/err.log || { echo "chunk $c failed"; tail -5 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/c${c}.json")); r=d["roofline"]
print("chunk %5d value %.0f  ms/step %.4f  dom %s %.4f ms shares %s" % ($c, d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["kernel_ms_share"]))
PY
done
