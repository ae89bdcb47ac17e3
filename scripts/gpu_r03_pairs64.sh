#!/bin/bash
# Round 3: fp64 engines on (re, im) pairs (default where the run kernels and the systolic MAC serve the engine) against the
# reference's groups of four (BFIR_F64_PAIRS=0): the whole GPU suite both ways, then the fp64 workloads, one box.
set -o pipefail
OUT=gpurun_out/${1:-r03an}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pairs64.txt || exit 1
BFIR_F64_PAIRS=0 timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee -a $OUT/pairs64.txt || exit 1
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
run() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" $tag | tee -a $OUT/pairs64.txt; }
for rep in 1 2; do for m in 1 0; do
  run cfg5_pairs$m BFIR_F64_PAIRS=$m -- --workload cfg5_2ch_262144tap_L4096_fp64
  for C in 2 3 8; do run plugin_f32frames_C${C}_pairs$m BFIR_F64_PAIRS=$m -- --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels $C; done
  run plugin_f64frames_C2_pairs$m BFIR_F64_PAIRS=$m -- --workload plugin_2ch_65536tap_L1024_fp64
done; done
