#!/bin/bash
# HBM-side traffic per launch of the three kernels (FETCH_SIZE / WRITE_SIZE in separate --pmc passes)
set -o pipefail
OUT=gpurun_out/${1:-traffic}; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --blocks 32768 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-exclusive-pass --no-kernel-events"
SHA=$(python -c "import bench; print(bench.csrc_sha())")
echo "{\"workload\": \"cfg3_8ch_131072tap_L4096_fp32\", \"chunk\": 4096, \"csrc_sha\": \"$SHA\", \"command\": \"$CMD\"}" > $OUT/pmc_meta.json
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- $CMD > $OUT/rocprof_fetch.log 2>&1 || { tail -20 $OUT/rocprof_fetch.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- $CMD > $OUT/rocprof_write.log 2>&1 || { tail -20 $OUT/rocprof_write.log; exit 1; }
python scripts/summarize_pmc.py $OUT 2>&1 | tee $OUT/pmc_summary.txt
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete
