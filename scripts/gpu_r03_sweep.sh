set -o pipefail
OUT=gpurun_out/r03e; mkdir -p $OUT
B="python bench.py --steps 6 --warmup 2 --no-cpu-timing --no-extras --blocks 65536"
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.4f" % r["pipeline"]["ms_per_launch_set"], "overlapped", {k: round(v["avg_launch_ms"],3) for k,v in r["kernels"].items()}, "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
timeout -k 10 300 python -m pytest tests/test_mac_sys_gpu.py -x -q 2>&1 | tail -2
BFIR_SYS_D=4 timeout -k 10 300 python -m pytest tests/test_mac_sys_gpu.py -x -q -k "headline or B20" 2>&1 | tail -1
BFIR_SYS_D=8 timeout -k 10 300 python -m pytest tests/test_mac_sys_gpu.py -x -q -k "headline or B20" 2>&1 | tail -1
BFIR_MAC_SYS=0 timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" stream | tee -a $OUT/ab.txt
for D in 4 8 12; do for W in 1024 1280 1536; do
  BFIR_MAC_SYS=1 BFIR_SYS_D=$D BFIR_SYS_WGS=$W timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" sys_D${D}_W$W | tee -a $OUT/ab.txt
done; done
