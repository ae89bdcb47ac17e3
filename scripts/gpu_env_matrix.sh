#!/bin/bash
# the whole -m gpu suite under the tuning switches that select other code paths (regression sweep)
OUT=gpurun_out/${1:-envmatrix}; mkdir -p $OUT
for sw in "BFIR_PAIR=0" "BFIR_DIRECT=0" "BFIR_PAIR_PERSIST=0" "BFIR_NO_MAC_SMALL=1" "BFIR_NO_BOUNCE=1" "BFIR_NO_SMALL_RUN=1" "BFIR_PIPE=1" "BFIR_MAC_BATCHED=1" "BFIR_MAC_SYS=0" "BFIR_MAC_SYS=1" "BFIR_PAIR_TIME=0" "BFIR_RUN64=0" "BFIR_F64_PAIRS=0" "BFIR_PIPE=3"; do
  tag=$(echo $sw | tr '=' '_')
  env $sw timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/$tag.log 2>&1; echo "$sw rc=$? $(tail -1 $OUT/$tag.log)"
done
