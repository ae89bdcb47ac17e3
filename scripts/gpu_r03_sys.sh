#!/bin/bash
# Round 3: the systolic MAC (csrc/mac_sys.hip) -- parity, then A/B timing against the streaming MAC on one box.
set -o pipefail
OUT=gpurun_out/${1:-r03b}
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity" | tee $OUT/progress.log
timeout -k 10 600 python -m pytest tests/test_mac_sys_gpu.py -x -q > $OUT/pytest.log 2>&1
rc=$?; tail -5 $OUT/pytest.log; echo "pytest rc=$rc" | tee -a $OUT/progress.log
[ $rc -ne 0 ] && exit $rc
B="python bench.py --steps 6 --warmup 2 --no-cpu-timing --no-extras --blocks 65536"
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.4f" % r["pipeline"]["ms_per_launch_set"], "overlapped", {k: round(v["avg_launch_ms"],3) for k,v in r["kernels"].items()}, "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()}, "parity", d.get("parity_rel_err_vs_oracle"))'
echo "== A/B" | tee -a $OUT/progress.log
for rep in 1 2; do
  BFIR_MAC_SYS=0 timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" stream | tee -a $OUT/ab.txt || exit 1
  BFIR_MAC_SYS=1 timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" sys | tee -a $OUT/ab.txt || exit 1
done
echo "== workgroups in flight (sys)" | tee -a $OUT/progress.log
for W in 768 1024 1536 2048 2560; do
  BFIR_MAC_SYS=1 BFIR_SYS_WGS=$W timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" sys_W$W | tee -a $OUT/ab.txt || exit 1
done
echo "== run length sweep (sys)" | tee -a $OUT/progress.log
for R in 64 128 256 512; do
  BFIR_MAC_SYS=1 BFIR_MAC_RANGE=$R timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" sys_R$R | tee -a $OUT/ab.txt || exit 1
done
echo "== done" | tee -a $OUT/progress.log
