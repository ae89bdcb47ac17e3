#!/bin/bash
# One GPU-box session: parity tests, bench, chunk sweep, rocprof kernel stats, PMC passes.
# Usage (from the repo root, through gpurun):  bash scripts/gpu_check.sh [tag] [chunk]
set -o pipefail
TAG=${1:-r01}
CHUNK=${2:-4096}
WORKLOAD=cfg3_8ch_131072tap_L4096_fp32
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/progress.log
tail -3 $OUT/pytest_gpu.log
echo "== bench default" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo bench failed; tail -20 $OUT/bench_default.err; exit 1; }
cat $OUT/bench_default.json
echo "== chunk sweep" | tee -a $OUT/progress.log
for c in ${SWEEP:-512 1024 2048}; do
  timeout -k 10 300 python bench.py --chunk $c --steps 4 --warmup 1 --no-cpu-baseline >> $OUT/bench_sweep.jsonl 2>> $OUT/bench_sweep.err || exit 1
done
python - <<PY
import json
for l in open("$OUT/bench_sweep.jsonl"):
    d=json.loads(l); r=d["roofline"]; print(d["config"]["blocks_per_launch"], d["value"], r["kernel_ms_share"], r["kernel"], r["avg_launch_ms"])
PY
echo "== rocprof kernel stats" | tee -a $OUT/progress.log
CMD="python bench.py --chunk $CHUNK --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive-pass"
echo "{\"workload\": \"$WORKLOAD\", \"chunk\": $CHUNK, \"command\": \"$CMD\"}" > $OUT/pmc_meta.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- $CMD > $OUT/rocprof_stats.log 2>&1 || { tail -20 $OUT/rocprof_stats.log; exit 1; }
echo "== rocprof pmc FETCH_SIZE" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- $CMD --no-kernel-events > $OUT/rocprof_fetch.log 2>&1 || { tail -20 $OUT/rocprof_fetch.log; exit 1; }
echo "== rocprof pmc WRITE_SIZE" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- $CMD --no-kernel-events > $OUT/rocprof_write.log 2>&1 || { tail -20 $OUT/rocprof_write.log; exit 1; }
python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
# keep only small files for the merge back
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete
echo "== done" | tee -a $OUT/progress.log
