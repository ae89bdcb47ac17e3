#!/bin/bash
# Round 3: k_mac_sys (DPP form) against the default MAC kernels, per workload, same box.
set -o pipefail
OUT=gpurun_out/${1:-r03i}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.4f" % r["pipeline"]["ms_per_launch_set"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in cfg3_8ch_131072tap_L4096_fp32 plugin_2ch_65536tap_L1024_fp64_f32frames cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp32 plugin_8ch_65536tap_L1024_fp32 cfg2_2ch_65536tap_L8192_fp32; do
for m in 0 1; do
  BFIR_MAC_SYS=$m timeout -k 10 300 python bench.py --workload $wl --blocks 65536 --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_sys$m | tee -a $OUT/sysab.txt
done; done
