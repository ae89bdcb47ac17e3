#!/bin/bash
# Round-3 evidence run (one GPU box): default bench line, rocprofv3 kernel stats of the same command, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, stamped with the csrc/ sha so bench.py can refuse a stale
# file), SQ counters of the three kernels, the other BASELINE configs, the plug-in's shape, the host path.
# Usage:  bash scripts/gpu_r02_profiles.sh [tag]
set -o pipefail
TAG=${1:-r03}
WORKLOAD=cfg3_8ch_131072tap_L4096_fp32
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SHA=$(python -c "import bench; print(bench.csrc_sha())")
echo "csrc sha $SHA" | tee $OUT/progress.log
echo "== bench default" | tee -a $OUT/progress.log
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo bench failed; tail -20 $OUT/bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print(d['value'], r['frac'], r['useful_frac'], r['literal_8d_x_peak'], r['valu_frac'], r['pipeline'], r['exclusive_launch_ms'])"
CMD="python bench.py --blocks 32768 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-exclusive-pass"
echo "{\"workload\": \"$WORKLOAD\", \"chunk\": 4096, \"csrc_sha\": \"$SHA\", \"command\": \"$CMD\"}" > $OUT/pmc_meta.json
echo "== rocprof kernel stats" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- $CMD > $OUT/rocprof_stats.log 2>&1 || { tail -20 $OUT/rocprof_stats.log; exit 1; }
cp $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
echo "== rocprof pmc FETCH_SIZE" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- $CMD --no-kernel-events > $OUT/rocprof_fetch.log 2>&1 || { tail -20 $OUT/rocprof_fetch.log; exit 1; }
echo "== rocprof pmc WRITE_SIZE" | tee -a $OUT/progress.log
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- $CMD --no-kernel-events > $OUT/rocprof_write.log 2>&1 || { tail -20 $OUT/rocprof_write.log; exit 1; }
python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
echo "== SQ counters (serial schedule: each kernel alone)" | tee -a $OUT/progress.log
SQCMD="python bench.py --blocks 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-exclusive-pass --no-kernel-events"
export BFIR_PIPE=1
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq1 -o p1 -- $SQCMD > $OUT/sq1.log 2>&1 || tail -5 $OUT/sq1.log
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -o p2 -- $SQCMD > $OUT/sq2.log 2>&1 || tail -5 $OUT/sq2.log
unset BFIR_PIPE
python - <<PY > $OUT/sq_summary.txt
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for n in ("k_mac","k_fwd","k_inv"):
            if n in k: k=n; break
        else: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# SQ counters per launch (4096 blocks of the headline shape), serial schedule (BFIR_PIPE=1), sums over the device")
for k in sorted(acc):
    c={n: sum(v)/len(v) for n,v in acc[k].items()}
    wc=c.get("SQ_WAVE_CYCLES",0) or 1; w=c.get("SQ_WAVES",0) or 1
    print("%-6s waves %7d  VALU insts/wave %6d  wave-cycles/wave (quad) %7d  VALU-active %5.1f%%  active-any %5.1f%%  wait-inst-any %5.1f%%  wait-any %5.1f%%  wait-inst-LDS %4.1f%%  VMEM rd/wr per wave %d/%d  LDS insts/wave %d  LDS bank-conflict cycles %d" % (
        k, w, c.get("SQ_INSTS_VALU",0)/w, wc/w, 100*c.get("SQ_ACTIVE_INST_VALU",0)/wc, 100*c.get("SQ_ACTIVE_INST_ANY",0)/wc,
        100*c.get("SQ_WAIT_INST_ANY",0)/wc, 100*c.get("SQ_WAIT_ANY",0)/wc, 100*c.get("SQ_WAIT_INST_LDS",0)/wc,
        c.get("SQ_INSTS_VMEM_RD",0)/w, c.get("SQ_INSTS_VMEM_WR",0)/w, c.get("SQ_INSTS_LDS",0)/w, c.get("SQ_LDS_BANK_CONFLICT",0)))
PY
cat $OUT/sq_summary.txt
echo "== other configs" | tee -a $OUT/progress.log
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" --steps 6 --warmup 2 --no-extras --no-cpu-timing > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return; }
  python - <<PY | tee -a $OUT/other_configs.txt
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("%-22s %-34s %9.1f Msamples/s  %8.3f ms/step  blocks/step %d  blocks/launch %d  streams %d  parity %.2e  exclusive ms %s" % (
    "$name", d["config"]["workload"], d["value"], d["ms_per_step"], d["config"]["blocks_per_step"],
    d["config"]["blocks_per_launch"], d["config"]["streams_total"], d["parity_rel_err_vs_oracle"] if d["parity_rel_err_vs_oracle"] is not None else -1, {k: round(v, 4) for k, v in (r.get("exclusive_launch_ms") or {}).items()}))
PY
}
run cfg2 --workload cfg2_2ch_65536tap_L8192_fp32
run cfg4_256streams --workload cfg4_stereo_65536tap_L4096_fp32 --streams 256
run cfg5_fp64 --workload cfg5_2ch_262144tap_L4096_fp64
BFIR_MAC_SYS=0 run cfg5_fp64_r02_mac --workload cfg5_2ch_262144tap_L4096_fp64
BFIR_RUN64=0 run cfg5_fp64_one_transform_kernels --workload cfg5_2ch_262144tap_L4096_fp64
run plugin_fp64_f32frames --workload plugin_2ch_65536tap_L1024_fp64_f32frames      # the plug-in as shipped
BFIR_MAC_SYS=0 run plugin_fp64_f32frames_r02_mac --workload plugin_2ch_65536tap_L1024_fp64_f32frames
run plugin_fp64_f32frames_chunk4096 --workload plugin_2ch_65536tap_L1024_fp64_f32frames --chunk 4096
run plugin_fp64_f64frames --workload plugin_2ch_65536tap_L1024_fp64
run plugin_fp64_f32frames_8ch --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels 8       # 7.1 audio through the shipped precision
BFIR_DIRECT=0 run plugin_fp64_f32frames_8ch_staging --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels 8
run plugin_fp64_f32frames_6ch --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels 6
for C in 5 3 1; do run plugin_fp64_f32frames_${C}ch --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels $C; done
BFIR_RUN64=0 run plugin_fp64_f32frames_one_transform_kernels --workload plugin_2ch_65536tap_L1024_fp64_f32frames
BFIR_RUN64=0 run plugin_fp64_f32frames_5ch_staging_r02 --workload plugin_2ch_65536tap_L1024_fp64_f32frames --channels 5
for T in 49152 98304 131072 196608 262144; do run plugin_fp64_f32frames_${T}taps --workload plugin_2ch_${T}tap_L1024_fp64_f32frames; done   # 48 ... 256 partitions
BFIR_MAC_SYS=0 run plugin_fp64_f32frames_262144taps_lds_mac --workload plugin_2ch_262144tap_L1024_fp64_f32frames
run plugin_fp32 --workload plugin_2ch_65536tap_L1024_fp32
run plugin_8ch_B48 --workload plugin_8ch_49152tap_L1024_fp32
run hl_8ch_24_partitions --workload hl_8ch_98304tap_L4096_fp32 --blocks 65536
run plugin_8ch_B64 --workload plugin_8ch_65536tap_L1024_fp32
run plugin_8ch_B128 --workload plugin_8ch_131072tap_L1024_fp32
BFIR_MAC_SYS=0 run plugin_8ch_B128_r02_mac --workload plugin_8ch_131072tap_L1024_fp32
BFIR_MAC_SYS=1 run cfg3_systolic_mac --workload cfg3_8ch_131072tap_L4096_fp32 --blocks 65536
BFIR_PAIR=0 run cfg3_staging_path --workload cfg3_8ch_131072tap_L4096_fp32 --blocks 65536
for C in 7 5 3 1; do run cfg3_${C}ch --workload cfg3_8ch_131072tap_L4096_fp32 --channels $C --blocks 65536; done
BFIR_PAIR_TIME=0 run cfg3_7ch_r02_path --workload cfg3_8ch_131072tap_L4096_fp32 --channels 7 --blocks 65536
echo "== plug-in shape + host path" | tee -a $OUT/progress.log
for g in 0.01 0.0005; do timeout -k 10 300 python scripts/plugin_shape.py $g 2>&1 | grep realsize; done | tee $OUT/plugin_shape.txt
timeout -k 10 300 python scripts/host_path.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_path.txt
bash scripts/gpu_lat_prof.sh $TAG/latprof 2>&1 | grep -v amdgpu.ids | tee $OUT/latency_kernels.txt
echo "== two ranks on this one GPU (gloo), both sharding modes" | tee -a $OUT/progress.log
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --device 0 --blocks 32768 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $OUT/gpus2_replicas.json 2>$OUT/gpus2.err; python -c "import json; d=json.load(open('$OUT/gpus2_replicas.json')); print('replicas', d['n_gpus'], d['scaling'], d['value'])"
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --device 0 --shard channels --blocks 32768 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $OUT/gpus2_channels.json 2>>$OUT/gpus2.err; python -c "import json; d=json.load(open('$OUT/gpus2_channels.json')); print('channels', d['n_gpus'], d['scaling'], d['value'])"
find $OUT -name "*.db" -delete; find $OUT -size +8M -delete
timeout -k 10 600 python bench.py --gpus 6 --dist-backend gloo --device 0 --shard channels --blocks 4096 --steps 4 --warmup 1 --no-extras --no-cpu-timing --no-exclusive-pass > $OUT/gpus6_channels.json 2>>$OUT/gpus2.err; python -c "import json; d=json.load(open('$OUT/gpus6_channels.json')); print('six ranks, channel shares 2 2 1 1 1 1', d['n_gpus'], d['scaling'], d['value'], d['parity_rel_err_vs_oracle'])"
echo "== done" | tee -a $OUT/progress.log
