#!/bin/bash
# Round 3: k_fwd_run / k_inv_run (fp64 direct mode, runs of blocks per workgroup) against the one-transform kernels
# (BFIR_RUN64=0) and over the run length, one box.
set -o pipefail
OUT=gpurun_out/${1:-r03ac}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in ${WLS:-cfg5_2ch_262144tap_L4096_fp64}; do
for len in ${LENS:-0 default 2 4 8 16 32}; do
  if [ $len = default ]; then unset BFIR_RUN64; else export BFIR_RUN64=$len; fi
  timeout -k 10 300 python bench.py --workload $wl ${EXTRA} --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_run$len | tee -a $OUT/run64.txt
done; done
