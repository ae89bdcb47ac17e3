#!/bin/bash
# chunk sweep at small launches (is X/Y cache residency worth anything with today's kernels?) + dither tests
set -o pipefail
OUT=gpurun_out/${1:-r02c}
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests/test_dither_gpu.py tests/test_formats_gpu.py tests/test_host_mirror_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for c in 64 128 256 512 1024 4096; do
  timeout -k 10 300 python bench.py --chunk $c --blocks 32768 --steps 6 --warmup 2 --no-cpu-baseline --no-extras >> $OUT/sweep.jsonl 2>> $OUT/sweep.err || exit 1
done
python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["config"]["blocks_per_launch"], d["value"], r["kernel_ms_share"], {k:v["avg_launch_ms"] for k,v in r["kernels"].items()}, r.get("exclusive_launch_ms"))
PY
