#!/bin/bash
# Round 3: SQ counters of the three kernels of fp64 engines (k_fwd_run, k_mac_sys, k_inv / k_inv_run), each alone on the GPU
# (fp64 engines run on one stream), for the plug-in's shape and cfg5.  Separate --pmc passes, kernel trace in its own run.
set -o pipefail
OUT=gpurun_out/${1:-r03sq64}; mkdir -p $OUT; export TMPDIR=/tmp
for wl in plugin_2ch_65536tap_L1024_fp64_f32frames cfg5_2ch_262144tap_L4096_fp64; do
  CMD="python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-exclusive-pass --no-kernel-events"
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${wl}_p1 -o p1 -- $CMD > $OUT/${wl}_p1.log 2>&1 || tail -5 $OUT/${wl}_p1.log
  timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/${wl}_p2 -o p2 -- $CMD > $OUT/${wl}_p2.log 2>&1 || tail -5 $OUT/${wl}_p2.log
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_kt -o kt -- $CMD > $OUT/${wl}_kt.log 2>&1 || tail -5 $OUT/${wl}_kt.log
  python - <<PY | tee -a $OUT/sq_fp64.txt
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${wl}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for n in ("k_mac_sys","k_fwd_run","k_inv_run","k_inv"):
            if n in k: k=n; break
        else: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur={}
for f in glob.glob("$OUT/${wl}_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for n in ("k_mac_sys","k_fwd_run","k_inv_run","k_inv"):
            if n in r["Name"]: dur[n]=float(r["AverageNs"]); break
print("# $wl: SQ counters per launch, sums over the device (one stream: each kernel has the GPU to itself)")
for k in sorted(acc):
    c={n: sum(v)/len(v) for n,v in acc[k].items()}
    wc=c.get("SQ_WAVE_CYCLES",0) or 1; w=c.get("SQ_WAVES",0) or 1
    print("%-10s launch %7.1f us  waves %6d  VALU insts/wave %6d  SALU/wave %5d  VALU-active %5.1f%%  active-any %5.1f%%  wait-inst-any %5.1f%%  wait-any %5.1f%%  wait-inst-LDS %4.1f%%  VMEM rd/wr per wave %d/%d  LDS insts/wave %d  LDS bank-conflict cycles %d" % (
        k, dur.get(k,0)/1e3, w, c.get("SQ_INSTS_VALU",0)/w, c.get("SQ_INSTS_SALU",0)/w, 100*c.get("SQ_ACTIVE_INST_VALU",0)/wc, 100*c.get("SQ_ACTIVE_INST_ANY",0)/wc,
        100*c.get("SQ_WAIT_INST_ANY",0)/wc, 100*c.get("SQ_WAIT_ANY",0)/wc, 100*c.get("SQ_WAIT_INST_LDS",0)/wc,
        c.get("SQ_INSTS_VMEM_RD",0)/w, c.get("SQ_INSTS_VMEM_WR",0)/w, c.get("SQ_INSTS_LDS",0)/w, c.get("SQ_LDS_BANK_CONFLICT",0)))
PY
done
find $OUT -name "*.db" -delete; find $OUT -size +4M -delete
