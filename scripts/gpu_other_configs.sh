#!/bin/bash
# BASELINE.json configs other than the headline, on one GPU.
set -o pipefail
OUT=gpurun_out/${1:-other}
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -k 10 600 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("%-6s %-34s %9.1f Msamples/s  %8.3f ms/step  blocks/step %d  blocks/launch %d  streams %d  shares %s" % (
    "$name", d["config"]["workload"], d["value"], d["ms_per_step"], d["config"]["blocks_per_step"],
    d["config"]["blocks_per_launch"], d["config"]["streams_total"], r["kernel_ms_share"]))
PY
}
run cfg2 --workload cfg2_2ch_65536tap_L8192_fp32 --blocks 8192 --chunk 0
run cfg4 --workload cfg4_stereo_65536tap_L4096_fp32 --streams 256 --blocks 64 --chunk 0
run cfg5 --workload cfg5_2ch_262144tap_L4096_fp64 --blocks 8192 --chunk 0
run cfg3g --workload cfg3_8ch_131072tap_L4096_fp32        # headline again, for the same box
BFIR_PAIR=0 run cfg3_general_path --workload cfg3_8ch_131072tap_L4096_fp32
