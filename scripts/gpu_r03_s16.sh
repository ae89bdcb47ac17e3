#!/bin/bash
# Sixteen-stage systolic MAC (129 ... 256 partitions) and twelve partitions per stage (24 / 48 / 96 / 192): bit-identity tests, then the plug-in's shape with long impulses on the
# new default against round 3's earlier path (LDS-tiled MAC on the groups of four: BFIR_MAC_SYS=0).
set -o pipefail
OUT=gpurun_out/s16
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_mac_sys_gpu.py tests/test_isa_audit.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" --steps 6 --warmup 2 --no-extras --no-cpu-timing > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }
  python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("%-34s %-44s %9.1f Msamples/s  blocks/launch %d  parity %.2e  exclusive ms %s" % (
    "$name", d["config"]["workload"], d["value"], d["config"]["blocks_per_launch"],
    d["parity_rel_err_vs_oracle"] if d["parity_rel_err_vs_oracle"] is not None else -1, {k: round(v, 4) for k, v in (r.get("exclusive_launch_ms") or {}).items()}))
PY
}
for W in plugin_2ch_49152tap_L1024_fp64_f32frames plugin_2ch_98304tap_L1024_fp64_f32frames plugin_2ch_131072tap_L1024_fp64_f32frames plugin_2ch_196608tap_L1024_fp64_f32frames plugin_2ch_262144tap_L1024_fp64_f32frames; do
  run ${W}_default --workload $W && BFIR_MAC_SYS=0 run ${W}_lds_mac --workload $W || exit 1
done
run plugin_8ch_49152tap_fp32_default --workload plugin_8ch_49152tap_L1024_fp32 && BFIR_MAC_SYS=0 run plugin_8ch_49152tap_fp32_lds_mac --workload plugin_8ch_49152tap_L1024_fp32
echo done
