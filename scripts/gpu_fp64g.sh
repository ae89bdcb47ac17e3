#!/bin/bash
# fp64 MAC: partition-streaming register kernel (BFIR_MAC64_VARIANT 12; the one-wave and three-wave builds of profiles/r02_fp64_mac.txt were removed again) against the LDS-tiled default (0)
set -o pipefail
OUT=gpurun_out/${1:-fp64g}; mkdir -p $OUT
for v in 12; do
  BFIR_MAC64_VARIANT=$v timeout -k 10 600 python -m pytest tests/test_launch_geometry_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "fp64 or 8-" > $OUT/pytest_v$v.log 2>&1; echo "variant $v pytest rc=$?"; tail -3 $OUT/pytest_v$v.log
done
for v in 0 12 0 12; do
  for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64_f32frames; do
    BFIR_MAC64_VARIANT=$v timeout -k 10 300 python bench.py --workload $wl --blocks 16384 --steps 6 --warmup 2 --no-cpu-timing --no-extras > $OUT/${wl}_$v.json 2>>$OUT/err.log || { echo "$wl v$v failed"; tail -3 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/${wl}_$v.json")); r=d["roofline"]
print("variant=%-2s %-42s value %.0f parity %s exclusive %s" % ("$v", "$wl", d["value"], d.get("parity_rel_err_vs_oracle"), r.get("exclusive_launch_ms")))
PY
  done
done
