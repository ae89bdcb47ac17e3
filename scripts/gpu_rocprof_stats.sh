#!/bin/bash
# rocprofv3 kernel-trace statistics of the default bench command (overlapped schedule only)
set -o pipefail
OUT=gpurun_out/${1:-stats}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive-pass"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- $CMD > $OUT/rocprof_stats.log 2>&1 || { tail -20 $OUT/rocprof_stats.log; exit 1; }
tail -2 $OUT/rocprof_stats.log
python scripts/summarize_pmc.py $OUT | head -8
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete
