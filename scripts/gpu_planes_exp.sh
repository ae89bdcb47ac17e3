#!/bin/bash
# plane-wise FFT kernels (half the LDS, 3-4 workgroups per CU): parity under the switch, then timing per variant
set -o pipefail
OUT=gpurun_out/${1:-planes}
mkdir -p $OUT
BFIR_PAIR_PLANES=6 timeout -k 10 600 python -m pytest tests/test_pair_path_gpu.py "tests/test_launch_geometry_gpu.py::test_large_launches_match_oracle_and_small_launches[cfg3_pair_4096]" -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest(planes=6) rc=$?"; tail -3 $OUT/pytest.log
for pl in 0 4 6 8; do
  BFIR_PAIR_PLANES=$pl timeout -k 10 300 python bench.py --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pl$pl.json 2>>$OUT/err.log || { echo "pl$pl failed"; tail -3 $OUT/err.log; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/pl$pl.json")); r=d["roofline"]
print("planes=%s value %.0f ms/set %.4f overlapped %s exclusive %s" % ("$pl", d["value"], r["pipeline"]["ms_per_launch_set"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms")))
PY
done
