#!/bin/bash
# Round 3: k_mac_sys with the stages of a bin 16 / S lanes apart (one v_mov_b32_dpp per exchanged register, stores of the lower
# stages dropped by the buffer range check).  Bit-exactness first, then systolic MAC forced / forbidden per workload on one box.
set -o pipefail
OUT=gpurun_out/${1:-r03t}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_mac_sys_gpu.py -q -x 2>&1 | tail -2 | tee $OUT/bank.txt || exit 1
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in ${WLS:-cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64_f32frames plugin_2ch_65536tap_L1024_fp32 plugin_8ch_65536tap_L1024_fp32 plugin_8ch_98304tap_L1024_fp32 plugin_8ch_131072tap_L1024_fp32 cfg4_stereo_65536tap_L4096_fp32 cfg3_8ch_131072tap_L4096_fp32}; do
for m in 1 0; do
  BFIR_MAC_SYS=$m timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_sys$m | tee -a $OUT/bank.txt
done; done
