#!/bin/bash
# parity of everything, then one-block latencies (latency path) for the three shapes and its kernel durations
OUT=gpurun_out/${1:-latchk}; mkdir -p $OUT
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/host_path.py 2>&1 | grep "latency path\|host-buffer"
for v in BFIR_NO_MAC_SMALL BFIR_NO_BOUNCE; do echo "== $v=1"; env $v=1 timeout -k 10 300 python scripts/host_path.py 2>&1 | grep "latency path"; done
bash scripts/gpu_lat_prof.sh $1/prof 2>&1 | grep -v amdgpu.ids
