#!/bin/bash
# Round 3: fp64 workloads, three-stream schedule against the serial one (BFIR_PIPE=1) and the two-stage one (BFIR_PIPE=2), one box.
set -o pipefail
OUT=gpurun_out/${1:-r03ao}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.3f" % r["pipeline"]["ms_per_launch_set"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()})'
run() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>$OUT/err.log | python -c "$pick" $tag | tee -a $OUT/pipe64.txt; }
for wl in plugin_2ch_65536tap_L1024_fp64_f32frames cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp32; do
  for p in 3 2 1; do run ${wl}_pipe$p BFIR_PIPE=$p -- --workload $wl; done
done
run plugin_f32frames_C8_pipe3 BFIR_PIPE=3 -- --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels 8
run plugin_f32frames_C8_pipe1 BFIR_PIPE=1 -- --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels 8
