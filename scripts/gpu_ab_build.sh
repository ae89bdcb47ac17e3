#!/bin/bash
# A/B two builds of the library on the GPU box: $1 = extra hipcc flags of variant B.
set -o pipefail
OUT=gpurun_out/${2:-ab}
mkdir -p $OUT
for rep in 1 2; do
  for v in A B; do
    if [ $v = B ]; then export BFIR_EXTRA_FLAGS="$1"; else export BFIR_EXTRA_FLAGS=""; fi
    python -c "import foo_dsp_bfir_amd as b; b.build(force=True)" > /dev/null 2>&1 || { echo build $v failed; exit 1; }
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/${v}_$rep.json 2>>$OUT/err.log || { tail -3 $OUT/err.log; exit 1; }
    python - <<PY
import json
d=json.load(open("$OUT/${v}_$rep.json")); r=d["roofline"]
print("$v rep $rep value", d["value"], "shares", r["kernel_ms_share"], "dom ms", r["avg_launch_ms"])
PY
  done
done
