#!/bin/bash
# Round 3: what would product spectra Y that never reach HBM be worth -- alone, and together with X?
# "alias" build = -DBFIR_EXPERIMENT_ALIAS -DBFIR_NT_X=0 -DBFIR_NT_Y=0 (plain cache policy; BFIR_X_ALIAS / BFIR_Y_ALIAS fold the
# delay line / the product spectra into that many slots; results garbage, instruction streams and launch geometry unchanged).
set -o pipefail
OUT=gpurun_out/${1:-r03g}; mkdir -p $OUT
run() { # name lib [env...]
  name=$1; lib=$2; shift 2
  if [ "$lib" = product ]; then unset BFIR_LIB_OVERRIDE; else export BFIR_LIB_OVERRIDE=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_$lib.so; fi
  env "$@" timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/$name.json 2>>$OUT/err.log || { echo "$name failed"; tail -3 $OUT/err.log; return; }
  python - <<PY | tee -a $OUT/alias.txt
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("%-18s value %.0f ms/set %.4f exclusive %s" % ("$name", d["value"], r["pipeline"]["ms_per_launch_set"], {k: round(v,3) for k,v in r.get("exclusive_launch_ms",{}).items()}))
PY
}
for rep in 1 2; do
run product product A=1
run plain alias A=1
run y_alias_64 alias BFIR_Y_ALIAS=64
run y_alias_256 alias BFIR_Y_ALIAS=256
run x_alias_128 alias BFIR_X_ALIAS=128
run xy_alias_128_64 alias BFIR_X_ALIAS=128 BFIR_Y_ALIAS=64
run xy_alias_512_256 alias BFIR_X_ALIAS=512 BFIR_Y_ALIAS=256
run sys_xy_alias_128_64 alias BFIR_X_ALIAS=128 BFIR_Y_ALIAS=64 BFIR_MAC_SYS=1
done
