#!/bin/bash
# time bench.py's exclusive (serial schedule) kernel durations for prebuilt library variants: $1 = tag, rest = variant names
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset BFIR_LIB_OVERRIDE; else export BFIR_LIB_OVERRIDE=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_$v.so; fi
  timeout -k 10 300 python bench.py --blocks ${BLOCKS:-32768} --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/$v.json 2>>$OUT/err.log || { echo "$v failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/$v.json")); r=d["roofline"]
print("%-10s value %.0f ms/set %.4f overlapped %s exclusive %s" % ("$v", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done
