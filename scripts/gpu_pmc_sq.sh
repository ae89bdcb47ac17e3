#!/bin/bash
# SQ counters for the kernels of one bench run (wave cycles, waits, active/issued instruction classes).
set -o pipefail
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --chunk ${CHUNK:-256} --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1 || { tail -20 $OUT/p1.log; }
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1 || { tail -20 $OUT/p2.log; }
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; }
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for n in ("k_mac","k_fwd","k_inv","k_stage_in","k_stage_out"):
            if n in k: k=n; break
        else: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(acc[k].items())})
PY
find $OUT -name "*.db" -delete; find $OUT -size +4M -delete
