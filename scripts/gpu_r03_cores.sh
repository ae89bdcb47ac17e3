#!/bin/bash
# Round 3: do the fp64 FFT kernels (LDS-bound occupancy: 2 workgroups of 150 registers per CU) and the systolic MAC (3 workgroups of
# 161-168 registers per CU = no room for anything else) share CUs better when the MAC leaves room?  Prefetch depth 8 = 169 registers
# = two MAC workgroups per CU; BFIR_SYS_WGS = MAC workgroups in flight.
set -o pipefail
OUT=gpurun_out/${1:-r03w}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.3f" % r["pipeline"]["ms_per_launch_set"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()})'
for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64_f32frames; do
for d in 6 8 4; do for w in 0 256 512 768; do
  BFIR_SYS_D=$d BFIR_SYS_WGS=$w timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_D${d}_wgs$w | tee -a $OUT/cores.txt
done; done; done
