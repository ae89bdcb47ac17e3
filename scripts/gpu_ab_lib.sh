#!/bin/bash
# A/B two prebuilt libraries in one call: A = lib/libbfir_hip.so, B = lib/libbfir_hip_B.so (BFIR_LIB_OVERRIDE)
set -o pipefail
OUT=gpurun_out/${1:-ablib}
mkdir -p $OUT
B=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_B.so
for rep in 1 2; do
  for C in ${CHUNKS:-256 512}; do
    for v in A B; do
      if [ $v = B ]; then export BFIR_LIB_OVERRIDE=$B; else unset BFIR_LIB_OVERRIDE; fi
      timeout -k 10 300 python bench.py --chunk $C --steps 6 --warmup 2 --blocks 2048 --no-cpu-baseline > $OUT/${v}_c${C}_$rep.json 2>>$OUT/err.log || { echo "$v failed"; tail -3 $OUT/err.log; continue; }
      python - <<PY
import json
d=json.load(open("$OUT/${v}_c${C}_$rep.json")); r=d["roofline"]
print("$v rep $rep chunk $C value %.0f shares %s" % (d["value"], r["kernel_ms_share"]))
PY
    done
  done
done
