#!/bin/bash
# k_fwd blocks-per-workgroup sweep (BFIR_FWD_RUN), serial schedule for clean kernel times and default schedule
set -o pipefail
OUT=gpurun_out/${1:-fwdrun}
mkdir -p $OUT
for C in ${CHUNKS:-256 512}; do
  for P in 1 3; do
    for R in ${RUNS:-1 2 4}; do
      BFIR_PIPE=$P BFIR_FWD_RUN=$R timeout -k 10 300 python bench.py --chunk $C --steps 6 --warmup 2 --blocks 2048 --no-cpu-baseline > $OUT/p${P}_r${R}_c$C.json 2>> $OUT/err.log || { echo "run $R chunk $C failed"; tail -5 $OUT/err.log; continue; }
      python - <<PY
import json
d=json.load(open("$OUT/p${P}_r${R}_c$C.json")); r=d["roofline"]
print("pipe $P run $R chunk $C value %.0f  ms/step %.4f  shares %s" % (d["value"], d["ms_per_step"], r["kernel_ms_share"]))
PY
    done
  done
done
