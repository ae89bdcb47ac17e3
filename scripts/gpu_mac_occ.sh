#!/bin/bash
# MAC exclusive time (serial schedule) for chunk / range pairs that give 2, 3 or 4 waves per SIMD
set -o pipefail
OUT=gpurun_out/${1:-macocc}
mkdir -p $OUT
for cr in ${PAIRS:-"256 64" "384 64" "512 128" "512 96" "768 128" "768 64" "1024 128"}; do
  set -- $cr; C=$1; R=$2
  for P in 1 3; do
    BFIR_PIPE=$P BFIR_MAC_RANGE=$R timeout -k 10 300 python bench.py --chunk $C --steps 4 --warmup 2 --blocks 3072 --no-cpu-baseline > $OUT/c${C}_r${R}_p$P.json 2>> $OUT/err.log || { echo "c $C r $R failed"; tail -5 $OUT/err.log; continue; }
    python - <<PY
import json
d=json.load(open("$OUT/c${C}_r${R}_p$P.json")); r=d["roofline"]; sh=r["kernel_ms_share"]
tot = d["ms_per_step"]
print("pipe $P chunk %4d range %3d value %.0f  mac share %.3f  mac us/256blk %.1f  (all kernels us/256blk %.1f)" % ($C, $R, d["value"], sh["k_mac"], r["avg_launch_ms"]*1000*256/$C if r["kernel"]=="k_mac" else -1, tot*1000*256/3072))
PY
  done
done
