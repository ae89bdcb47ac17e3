#!/bin/bash
# Round 3: fp64 engines on float64 frames, channel pairs per workgroup in the direct-mode FFT kernels (whole frames per lane).
set -o pipefail
OUT=gpurun_out/${1:-r03y}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $OUT/f64pairs.txt || exit 1
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64 plugin_2ch_65536tap_L1024_fp64_f32frames; do
for rep in 1 2; do
  timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl} | tee -a $OUT/f64pairs.txt
done; done
for C in 4 8; do
  timeout -k 10 300 python bench.py --workload plugin_2ch_65536tap_L1024_fp64 --channels $C --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" plugin_fp64_f64frames_C$C | tee -a $OUT/f64pairs.txt
done
