#!/bin/bash
# fp32 with 17 ... 24 partitions: k_mac_sys (two stages of twelve) against k_mac_stream (a register batch of 32)
set -o pipefail
OUT=gpurun_out/b24
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" --steps 6 --warmup 2 --no-extras --no-cpu-timing > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }
  python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("%-34s %-44s %9.1f Msamples/s  blocks/launch %d  parity %.2e  exclusive ms %s" % (
    "$name", d["config"]["workload"], d["value"], d["config"]["blocks_per_launch"],
    d["parity_rel_err_vs_oracle"] if d["parity_rel_err_vs_oracle"] is not None else -1, {k: round(v, 4) for k, v in (r.get("exclusive_launch_ms") or {}).items()}))
PY
}
W=hl_8ch_98304tap_L4096_fp32
for i in 1 2; do
BFIR_MAC_SYS=1 run ${W}_sys_$i --workload $W --blocks 65536 && BFIR_MAC_SYS=0 run ${W}_stream_$i --workload $W --blocks 65536 || exit 1
done
echo done
