#!/bin/bash
# fp64 engines: parity, then cfg5 and the plug-in's shape per MAC variant (BFIR_MAC64_VARIANT: 0 = two bins per lane
# (new default), 6 = four bins per lane (round 1), 4 / 5 = other prefetch depths)
set -o pipefail
OUT=gpurun_out/${1:-fp64}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_launch_geometry_gpu.py -m gpu -x -q -k "fp64 or 8- or 8]" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in 0 6 4 5; do
  for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64; do
    BFIR_MAC64_VARIANT=$v timeout -k 10 300 python bench.py --workload $wl --blocks 8192 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${wl}_$v.json 2>>$OUT/err.log || { echo "$wl v$v failed"; tail -3 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/${wl}_$v.json")); r=d["roofline"]
print("variant=%s %-34s value %.0f shares %s exclusive %s" % ("$v", "$wl", d["value"], r["kernel_ms_share"], r.get("exclusive_launch_ms")))
PY
  done
done
