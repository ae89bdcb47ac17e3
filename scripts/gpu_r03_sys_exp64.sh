#!/bin/bash
# Round 3: what is a slot of the fp64 k_mac_sys (S = 4, 16 partitions per lane) made of?  Experiment builds of mac_sys.hip
# (-DBFIR_SYS_EXP bits, results garbage; built in the container with _build.build_variant("e<bits>", ["-DBFIR_SYS_EXP=<bits>"],
# only=["mac_sys.hip"])): 1 no partial-sum hand-over, 2 no relay of x, 4 no stores, 8 no loads, 32 no tail test.
set -o pipefail
OUT=gpurun_out/${1:-r03s}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()})'
L=$PWD/foo-dsp-bfir_amd/lib
for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64_f32frames; do
for v in "" e1 e2 e3 e4 e8 e15 e47; do
  if [ -n "$v" ]; then export BFIR_LIB_OVERRIDE=$L/libbfir_hip_$v.so; else unset BFIR_LIB_OVERRIDE; fi
  timeout -k 10 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>$OUT/err.log | python -c "$pick" "${wl}_${v:-product}" | tee -a $OUT/exp64.txt
done; done
