#!/bin/bash
# Build with -DBFIR_TRACE into the in-tree lib ON THE GPU BOX (the snapshot there is scratch), print phase timings.
set -o pipefail
mkdir -p gpurun_out
BFIR_EXTRA_FLAGS="-DBFIR_TRACE" python -c "import foo_dsp_bfir_amd as b; b.build(force=True)" > gpurun_out/trace_build.log 2>&1 || { tail -20 gpurun_out/trace_build.log; exit 1; }
timeout -k 10 300 python scripts/trace_phases.py ${1:-256} | tee gpurun_out/trace_${1:-256}.txt
