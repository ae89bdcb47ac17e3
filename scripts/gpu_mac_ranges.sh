#!/bin/bash
# Streaming-MAC range sweep (BFIR_MAC_RANGE blocks per wave) against the LDS kernel (variant 8).
set -o pipefail
OUT=gpurun_out/${1:-macr}
mkdir -p $OUT
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --chunk $C --steps 6 --warmup 2 --no-cpu-baseline > $OUT/${name}_c$C.json 2>> $OUT/err.log || { echo "$name chunk $C failed"; tail -5 $OUT/err.log; return; }
  python - <<PY
import json
d=json.load(open("$OUT/${name}_c$C.json")); r=d["roofline"]
print("%-10s chunk $C value %.0f  ms/step %.4f  dom %s %.4f ms shares %s" % ("$name", d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["kernel_ms_share"]))
PY
}
for C in ${CHUNKS:-256 512}; do
  run lds BFIR_MAC_VARIANT=8
  for R in ${RANGES:-32 64 128}; do run stream$R BFIR_MAC_RANGE=$R; done
  run auto BFIR_MAC_VARIANT=0
done
