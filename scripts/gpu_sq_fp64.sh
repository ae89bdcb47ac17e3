#!/bin/bash
# SQ counters + effective clock of the fp64 MAC kernels (LDS-tiled default vs partition-streaming), plug-in shape, serial schedule
set -o pipefail
OUT=gpurun_out/${1:-sq64}; mkdir -p $OUT; export TMPDIR=/tmp
export BFIR_PIPE=1
for v in 0 12; do
  export BFIR_MAC64_VARIANT=$v
  CMD="python bench.py --workload plugin_2ch_65536tap_L1024_fp64_f32frames --blocks 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-exclusive-pass --no-kernel-events"
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/v${v}_p1 -o p1 -- $CMD > $OUT/v${v}_p1.log 2>&1 || tail -5 $OUT/v${v}_p1.log
  timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/v${v}_p2 -o p2 -- $CMD > $OUT/v${v}_p2.log 2>&1 || tail -5 $OUT/v${v}_p2.log
  timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/v${v}_p3 -o p3 -- $CMD > $OUT/v${v}_p3.log 2>&1 || tail -5 $OUT/v${v}_p3.log
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v${v}_kt -o kt -- $CMD > $OUT/v${v}_kt.log 2>&1 || tail -5 $OUT/v${v}_kt.log
  python - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/v${v}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mac" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c={n: sum(x)/len(x) for n,x in acc.items()}
dur=None
for f in glob.glob("$OUT/v${v}_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mac" in r["Name"]: dur=float(r["AverageNs"])
wc=c.get("SQ_WAVE_CYCLES",1); w=c.get("SQ_WAVES",1)
print("variant $v: launch %.1f us  waves %d  VALU insts/wave %d  wave-cycles/wave(quad) %d  VALU-active %.1f%%  active-any %.1f%%  wait-inst-any %.1f%%  wait-any %.1f%%  wait-inst-LDS %.1f%%  VMEM-active %.1f%%  LDS/wave %d  VMEM rd/wave %d  L2 hit %.1f%%  FETCH %.1f MB  clock %.2f GHz" % (
  (dur or 0)/1e3, w, c.get("SQ_INSTS_VALU",0)/w, wc/w, 100*c.get("SQ_ACTIVE_INST_VALU",0)/wc, 100*c.get("SQ_ACTIVE_INST_ANY",0)/wc, 100*c.get("SQ_WAIT_INST_ANY",0)/wc, 100*c.get("SQ_WAIT_ANY",0)/wc,
  100*c.get("SQ_WAIT_INST_LDS",0)/wc, 100*c.get("SQ_ACTIVE_INST_VMEM",0)/wc, c.get("SQ_INSTS_LDS",0)/w, c.get("SQ_INSTS_VMEM_RD",0)/w,
  100*c.get("TCC_HIT_sum",0)/max(1,c.get("TCC_HIT_sum",0)+c.get("TCC_MISS_sum",0)), c.get("FETCH_SIZE",0)*2/1024, (c.get("GRBM_GUI_ACTIVE",0)/8/(dur or 1))))
PY
done
find $OUT -name "*.db" -delete; find $OUT -size +4M -delete
