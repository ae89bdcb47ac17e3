#!/bin/bash
# the plug-in's exact shape (fp64 arithmetic, float32 stereo frames): per-kernel times, stereo direct mode vs staging kernels
set -o pipefail
OUT=gpurun_out/${1:-fp64d}
mkdir -p $OUT
for v in 7 8; do
  BFIR_MAC64_VARIANT=$v timeout -k 10 600 python -m pytest tests/test_launch_geometry_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "fp64 or 8-" > $OUT/pytest_v$v.log 2>&1; echo "variant $v pytest rc=$?"; tail -2 $OUT/pytest_v$v.log
done
for d in default 0 1; do for v in 0 7; do
  if [ $d = default ]; then unset BFIR_DIRECT; else export BFIR_DIRECT=$d; fi
  BFIR_MAC64_VARIANT=$v timeout -k 10 300 python bench.py --workload plugin_2ch_65536tap_L1024_fp64_f32frames --blocks 32768 --steps 4 --warmup 1 --no-cpu-timing --no-extras > $OUT/p_${d}_$v.json 2>>$OUT/err.log || { echo "direct=$d v$v failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/p_${d}_$v.json")); r=d["roofline"]
print("direct=%s mac64=%s value %.0f parity %s overlapped %s exclusive %s" % ("$d", "$v", d["value"], d.get("parity_rel_err_vs_oracle"), {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done; done
