set -o pipefail
OUT=gpurun_out/r03l; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.4f" % r["pipeline"]["ms_per_launch_set"], "chunk", d["config"]["blocks_per_launch"], "blocks", d["config"]["blocks_per_step"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in plugin_2ch_65536tap_L1024_fp64_f32frames plugin_2ch_65536tap_L1024_fp64 cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp32 cfg4_stereo_65536tap_L4096_fp32; do
  extra=""; [ $wl = cfg4_stereo_65536tap_L4096_fp32 ] && extra="--streams 256"
  timeout -k 10 300 python bench.py --workload $wl $extra --steps 6 --warmup 2 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" $wl | tee -a $OUT/f64.txt
done
