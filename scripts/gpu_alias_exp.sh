#!/bin/bash
# Timing experiment: how fast would today's three kernels run if X and Y never left the caches?
# (library built with -DBFIR_EXPERIMENT_ALIAS -- _build.build_variant("alias", ["-DBFIR_EXPERIMENT_ALIAS"]) -- folds the
# delay line / product spectra into a few slots; results are garbage, instruction streams and launch geometry unchanged)
set -o pipefail
OUT=gpurun_out/${1:-alias}
mkdir -p $OUT
export BFIR_LIB_OVERRIDE=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_alias.so
run() {
  timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/$1.json 2>>$OUT/err.log || { echo "$1 failed"; tail -3 $OUT/err.log; return; }
  python - <<PY
import json
d=json.load(open("$OUT/$1.json")); r=d["roofline"]
print("%-14s value %.0f ms/set %.4f overlapped %s exclusive %s" % ("$1", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
}
run base
BFIR_X_ALIAS=128 BFIR_Y_ALIAS=64 run xy_alias
BFIR_X_ALIAS=128 run x_alias
BFIR_Y_ALIAS=64 run y_alias
BFIR_X_ALIAS=64 BFIR_Y_ALIAS=32 run xy_alias_64_32
BFIR_X_ALIAS=512 BFIR_Y_ALIAS=512 run xy_alias_512
run base2
