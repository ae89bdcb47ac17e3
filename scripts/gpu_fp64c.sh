#!/bin/bash
# fp64 engines after the round-2 FFT addressing / stereo direct mode / grouped-barrier MAC changes:
# parity first, then cfg5 and the plug-in's shape per MAC variant (BFIR_MAC64_VARIANT 0 = barrier per partition,
# 7 / 8 = per two / four partitions) and the plug-in's exact shape (float32 frames) with and without direct mode.
set -o pipefail
OUT=gpurun_out/${1:-fp64c}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_launch_geometry_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
for v in 7 8; do
  BFIR_MAC64_VARIANT=$v timeout -k 10 600 python -m pytest tests/test_launch_geometry_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "fp64 or 8-" > $OUT/pytest_v$v.log 2>&1; echo "variant $v pytest rc=$?"; tail -2 $OUT/pytest_v$v.log
done
for v in 0 7 8; do
  for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64; do
    BFIR_MAC64_VARIANT=$v timeout -k 10 300 python bench.py --workload $wl --blocks 16384 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${wl}_$v.json 2>>$OUT/err.log || { echo "$wl v$v failed"; tail -3 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/${wl}_$v.json")); r=d["roofline"]
print("variant=%s %-34s value %.0f shares %s exclusive %s" % ("$v", "$wl", d["value"], r["kernel_ms_share"], r.get("exclusive_launch_ms")))
PY
  done
done
for v in 0 7 8; do for d in 1 0; do echo "BFIR_MAC64_VARIANT=$v BFIR_DIRECT=$d"; BFIR_MAC64_VARIANT=$v BFIR_DIRECT=$d timeout -k 10 300 python scripts/plugin_shape.py 2>&1 | grep "realsize 8"; done; done
timeout -k 10 300 python scripts/host_path.py 2>&1 | grep "one run" 
