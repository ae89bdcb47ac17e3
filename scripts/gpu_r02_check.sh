#!/bin/bash
# Round-2 GPU session: new launch-geometry parity tests first, then the whole -m gpu suite, then bench.
set -o pipefail
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== geometry tests" | tee $OUT/progress.log
timeout -k 10 900 python -m pytest tests/test_launch_geometry_gpu.py tests/test_bench_spawn.py -m gpu -x -q -s > $OUT/pytest_geom.log 2>&1
echo "geom rc=$?" | tee -a $OUT/progress.log
tail -15 $OUT/pytest_geom.log
echo "== pytest -m gpu (all)" | tee -a $OUT/progress.log
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/progress.log
tail -3 $OUT/pytest_gpu.log
echo "== bench default" | tee -a $OUT/progress.log
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo bench failed; tail -20 $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default.json
echo "== done" | tee -a $OUT/progress.log
