#!/bin/bash
# A/B of the product library against a variant build (lib/libbfir_hip_<name>.so, made in the container with _build.build_variant)
# on one box: scripts/gpu_r03_ab.sh <outdir> <variant> <workload> [<workload> ...]
set -o pipefail
OUT=gpurun_out/$1; VAR=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_$2.so; shift 2; mkdir -p $OUT; test -f $VAR || exit 1
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in "$@"; do for rep in 1 2; do
  timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_product | tee -a $OUT/ab.txt
  BFIR_LIB_OVERRIDE=$VAR timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_variant | tee -a $OUT/ab.txt
done; done
