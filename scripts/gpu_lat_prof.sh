#!/bin/bash
# kernel durations inside the one-block latency path (rocprofv3 kernel trace), plug-in shape fp64 / fp32 and the headline shape
OUT=gpurun_out/${1:-latprof}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1024 64 2 8" "1024 64 2 4" "4096 32 8 4"; do
  tag=$(echo $cfg | tr ' ' '_')
  python3 $R/scripts/latency_probe.py $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$tag -o p -- python3 $R/scripts/latency_probe.py $cfg 200 > $R/$OUT/$tag.log 2>&1
  f=$(find $R/$OUT/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:8]: print("  %-60s calls %6s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
