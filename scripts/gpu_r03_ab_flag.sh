#!/bin/bash
# A/B on one box: this tree's library against an older build placed at foo-dsp-bfir_amd/lib/ab_old/libbfir_hip.so (built from a
# git worktree of the older commit; not committed), alternating runs of the headline bench (BFIR_LIB_OVERRIDE selects the library).
set -o pipefail
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export BFIR_LIB_OVERRIDE=$PWD/foo-dsp-bfir_amd/lib/ab_old/libbfir_hip.so; else unset BFIR_LIB_OVERRIDE; fi
    timeout -k 10 300 python bench.py --blocks 65536 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/ab/${v}_$i.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/ab/${v}_$i.json')); r=d['roofline']; print('$v', $i, d['value'], r['ms_per_launch_set'], r['exclusive_launch_ms'])"
  done
done
