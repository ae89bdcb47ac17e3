#!/bin/bash
# fp32, more than 32 partitions: partition-streaming register MAC (BFIR_MAC_TS=1: 24 outputs, 3 waves/SIMD; 2: 32 outputs, queue 8; 3: queue 16) vs the LDS MAC (0)
set -o pipefail
OUT=gpurun_out/${1:-ts32}; mkdir -p $OUT
for v in 1 2 3; do
  BFIR_MAC_TS=$v timeout -k 10 600 python -m pytest tests/test_launch_geometry_gpu.py -m gpu -x -q -k "plugin" > $OUT/pytest_v$v.log 2>&1; echo "ts $v pytest rc=$?"; tail -1 $OUT/pytest_v$v.log
done
for v in 0 1 2 3 0; do
  for wl in plugin_8ch_65536tap_L1024_fp32 plugin_8ch_131072tap_L1024_fp32 plugin_2ch_65536tap_L1024_fp32; do
    BFIR_MAC_TS=$v timeout -k 10 300 python bench.py --workload $wl --blocks 32768 --steps 6 --warmup 2 --no-cpu-timing --no-extras > $OUT/${wl}_$v.json 2>>$OUT/err.log || { echo "$wl v$v failed"; tail -3 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/${wl}_$v.json")); r=d["roofline"]
print("ts=%s %-34s value %.0f parity %.2e exclusive %s" % ("$v", "$wl", d["value"], d.get("parity_rel_err_vs_oracle") or -1, r.get("exclusive_launch_ms")))
PY
  done
done
