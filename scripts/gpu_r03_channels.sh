#!/bin/bash
# Round 3: the headline shape with 1 ... 8 channels.  Odd counts (and one channel) take the float
# fast path with blocks paired in time; BFIR_PAIR_TIME=0 = round 2's behaviour (general / direct path) for comparison.
set -o pipefail
OUT=gpurun_out/${1:-r03k}; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; C=d["config"]["channels"]; print(sys.argv[1], "value %.0f" % d["value"], "per channel %.0f" % (d["value"]/C), "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for C in 8 7 6 5 4 3 2 1; do
  timeout -k 10 300 python bench.py --channels $C --blocks 32768 --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" C$C | tee -a $OUT/channels.txt
done
for C in 7 5 3 1; do
  BFIR_PAIR_TIME=0 timeout -k 10 300 python bench.py --channels $C --blocks 32768 --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" C${C}_round2_path | tee -a $OUT/channels.txt
done
