set -o pipefail
OUT=gpurun_out/r03m; mkdir -p $OUT
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.4f" % r["pipeline"]["ms_per_launch_set"], "chunk", d["config"]["blocks_per_launch"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
for wl in cfg5_2ch_262144tap_L4096_fp64 plugin_2ch_65536tap_L1024_fp64_f32frames; do
for D in 4 6 8; do for W in 512 768 1024; do
  BFIR_SYS_D=$D BFIR_SYS_WGS=$W timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-timing --no-extras 2>$OUT/err.log | python -c "$pick" ${wl}_D${D}_W$W | tee -a $OUT/f64d.txt
done; done; done
