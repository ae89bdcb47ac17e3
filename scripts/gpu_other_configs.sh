#!/bin/bash
# Bench lines of the non-headline BASELINE configs (parity-test cases; recorded for DESIGN.md only).
set -o pipefail
TAG=${1:-cfgs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --workload cfg2_2ch_65536tap_L8192_fp32 --blocks 1024 --chunk 512 --steps 4 > $OUT/cfg2.json 2>$OUT/cfg2.err || tail -3 $OUT/cfg2.err
python bench.py --workload cfg5_2ch_262144tap_L4096_fp64 --blocks 1024 --chunk 256 --steps 4 > $OUT/cfg5.json 2>$OUT/cfg5.err || tail -3 $OUT/cfg5.err
python bench.py --workload cfg4_stereo_65536tap_L4096_fp32 --streams 32 --blocks 64 --chunk 32 --steps 4 --no-cpu-baseline > $OUT/cfg4_32streams.json 2>$OUT/cfg4.err || tail -3 $OUT/cfg4.err
python bench.py --workload cfg4_stereo_65536tap_L4096_fp32 --streams 256 --blocks 32 --chunk 16 --steps 4 --no-cpu-baseline > $OUT/cfg4_256streams.json 2>$OUT/cfg4b.err || tail -3 $OUT/cfg4b.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "value", d["value"], "Msamples/s; pct_alg_roofline", d["pct_of_hbm_roofline_algorithmic"], "dom", r["kernel"], r["kernel_ms_share"], "cpu", (d["cpu_baseline"] or {}).get("value"), "parity", d["parity_rel_err_vs_oracle"])
    except Exception as e: print(f, "ERR", e)
PY
