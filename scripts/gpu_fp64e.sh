#!/bin/bash
OUT=gpurun_out/${1:-fp64e}; mkdir -p $OUT
for g in 0.01 0.0005; do for d in default 0; do
  if [ $d = default ]; then unset BFIR_DIRECT; else export BFIR_DIRECT=$d; fi
  echo "BFIR_DIRECT=$d gain $g"; timeout -k 10 300 python scripts/plugin_shape.py $g 2>&1 | grep realsize
done; done
