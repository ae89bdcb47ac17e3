#!/bin/bash
# Round 3: the fp64 configurations after the run kernels became the default (forward: every fp64 engine of 1024 ... 8192 points in
# direct mode; inverse: where channel pairs per workgroup do not apply), next to BFIR_RUN64=0, one box.
set -o pipefail
OUT=gpurun_out/${1:-r03ai}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $OUT/f64final.txt || exit 1
BFIR_RUN64=0 timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee -a $OUT/f64final.txt || exit 1
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
run() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" $tag | tee -a $OUT/f64final.txt; }
for m in 1 0; do
  if [ $m = 1 ]; then E=X=1; T=run; else E=BFIR_RUN64=0; T=run0; fi
  run cfg5_$T $E -- --workload cfg5_2ch_262144tap_L4096_fp64
  for C in 1 2 3 4 5 6 7 8; do run plugin_f32frames_C${C}_$T $E -- --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels $C; done
  for C in 1 2 3 4 8; do run plugin_f64frames_C${C}_$T $E -- --workload plugin_2ch_65536tap_L1024_fp64 --channels $C; done
done
