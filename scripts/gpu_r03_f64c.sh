set -o pipefail
OUT=gpurun_out/r03o; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; C=d["config"]["channels"]; print(sys.argv[1], "value %.0f" % d["value"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for C in 2 3 4 6 8; do
  timeout -k 10 300 python bench.py --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels $C --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" plugin_fp64_f32frames_C$C | tee -a $OUT/f64c.txt
done
for C in 4 8; do
  timeout -k 10 300 python bench.py --workload plugin_2ch_65536tap_L1024_fp64 --channels $C --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" plugin_fp64_f64frames_C$C | tee -a $OUT/f64c.txt
done
