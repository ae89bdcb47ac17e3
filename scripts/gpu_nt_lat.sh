#!/bin/bash
# does the nontemporal policy of the X / Y streams cost anything on small launches?  (latency path, small chunks)
OUT=gpurun_out/${1:-ntlat}; mkdir -p $OUT
for v in base nt_all base nt_all; do
  if [ "$v" = base ]; then unset BFIR_LIB_OVERRIDE; else export BFIR_LIB_OVERRIDE=$PWD/foo-dsp-bfir_amd/lib/libbfir_hip_$v.so; fi
  echo "== $v"; timeout -k 10 300 python scripts/host_path.py 2>&1 | grep "latency path\|host-buffer"
  for c in 64 256; do
    timeout -k 10 300 python bench.py --blocks 8192 --chunk $c --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-exclusive-pass > $OUT/${v}_$c.json 2>>$OUT/err.log && python -c "
import json; d=json.load(open('$OUT/${v}_$c.json')); print('chunk $c value %.0f' % d['value'])"
  done
done
