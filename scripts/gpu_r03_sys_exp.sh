#!/bin/bash
# Round 3: what is a slot of k_mac_sys made of?  Experiment builds (-DBFIR_SYS_EXP bits, results garbage): 1 no partial-sum
# hand-over, 2 no relay of x, 4 no stores, 8 no loads, 16 no negated operand, 32 no tail test; exclusive kernel times.
set -o pipefail
OUT=gpurun_out/${1:-r03f}; mkdir -p $OUT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --blocks 16384"
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()})'
L=$PWD/foo-dsp-bfir_amd/lib
BFIR_MAC_SYS=0 timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" stream | tee -a $OUT/exp.txt
for v in "" ${VARIANTS:-e3 e4 e8 e16 e15 e31 e47 e63}; do
  if [ -n "$v" ]; then export BFIR_LIB_OVERRIDE=$L/libbfir_hip_$v.so; else unset BFIR_LIB_OVERRIDE; fi
  BFIR_MAC_SYS=1 timeout -k 10 300 $B 2>$OUT/err.log | python -c "$pick" "sys_${v:-product}" | tee -a $OUT/exp.txt
done
