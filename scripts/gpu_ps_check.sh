#!/bin/bash
# persistent pair kernels: parity (pair-path, launch-geometry, engine, fullsize suites), then A/B timing via the env switch
set -o pipefail
OUT=gpurun_out/${1:-ps}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_pair_path_gpu.py tests/test_launch_geometry_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_golden_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for v in 0 2 1; do
  BFIR_PAIR_PERSIST=$v timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/p$v.json 2>>$OUT/err.log || { echo "p$v failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/p$v.json")); r=d["roofline"]
print("persist=%s value %.0f ms/set %.4f overlapped %s exclusive %s" % ("$v", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done
