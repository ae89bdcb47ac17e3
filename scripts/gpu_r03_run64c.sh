#!/bin/bash
# Round 3: even channel counts -- two channels per workgroup (k_fwd / k_inv, CPW = 2) against the run kernels, per direction
# (BFIR_STEREO_WG: 1 both pair kernels, 0 both run kernels, 2 pair forward + run inverse, 3 run forward + pair inverse), one box.
set -o pipefail
OUT=gpurun_out/${1:-r03ag}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
run() { local tag=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" $tag | tee -a $OUT/run64c.txt; }
for wl in plugin_2ch_65536tap_L1024_fp64_f32frames plugin_2ch_65536tap_L1024_fp64; do for C in ${CS:-2 4 8}; do for m in 1 3 2 0; do for len in ${LENS:-8}; do
  run ${wl}_C${C}_stereo${m}_len$len BFIR_STEREO_WG=$m BFIR_RUN64=$len -- --workload $wl --channels $C
done; done; done; done
