#!/bin/bash
# full -m gpu suite, then the plug-in's shape with and without clipping output, then the nt policy variants
set -o pipefail
OUT=gpurun_out/${1:-chk2}; mkdir -p $OUT
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for g in 0.01 0.0005; do echo "gain $g"; timeout -k 10 300 python scripts/plugin_shape.py $g 2>&1 | grep realsize; done
