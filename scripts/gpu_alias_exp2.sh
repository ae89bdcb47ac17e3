#!/bin/bash
# Ceiling of cache-resident X / Y with the final kernels: product build vs the same kernels with the plain cache policy
# ("plain") vs plain + spectra folded into a few slots ("alias" = -DBFIR_EXPERIMENT_ALIAS -DBFIR_NT_X=0 -DBFIR_NT_Y=0).
set -o pipefail
OUT=gpurun_out/${1:-alias4}; mkdir -p $OUT
run() { # name lib [env...]
  name=$1; lib=$2; shift 2
  if [ "$lib" = product ]; then unset BFIR_LIB_OVERRIDE; else export BFIR_LIB_OVERRIDE=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_$lib.so; fi
  env "$@" timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/$name.json 2>>$OUT/err.log || { echo "$name failed"; tail -3 $OUT/err.log; return; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("%-16s value %.0f ms/set %.4f exclusive %s" % ("$name", d["value"], r["pipeline"]["ms_per_launch_set"], r.get("exclusive_launch_ms")))
PY
}
run product product A=1
run plain plain A=1
run alias_128_64 alias BFIR_X_ALIAS=128 BFIR_Y_ALIAS=64
run alias_64_32 alias BFIR_X_ALIAS=64 BFIR_Y_ALIAS=32
run product2 product A=1
run plain2 plain A=1
