#!/bin/bash
# A/B the pair path (two channels per transform on raw frames) against the planar staging path
set -o pipefail
OUT=gpurun_out/${1:-pairab}
mkdir -p $OUT
for C in ${CHUNKS:-256 512 1024}; do
  for P in 0 1; do
    BFIR_PAIR=$P timeout -k 10 300 python bench.py --chunk $C --steps 6 --warmup 2 --no-cpu-baseline > $OUT/pair${P}_c$C.json 2>> $OUT/err.log || { echo "pair $P chunk $C failed"; tail -5 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/pair${P}_c$C.json")); r=d["roofline"]
print("pair $P chunk $C value %.0f  ms/step %.4f  dom %s %.4f ms shares %s" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["kernel_ms_share"]))
PY
  done
done
