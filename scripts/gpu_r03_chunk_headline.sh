set -o pipefail
mkdir -p gpurun_out/r03aw
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "ms/set %.3f" % r["pipeline"]["ms_per_launch_set"], "chunk", d["config"]["blocks_per_launch"])'
for c in 4096 2048 8192 4096 3072 6144; do
  timeout -k 10 300 python bench.py --chunk $c --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-exclusive-pass 2>/dev/null | python -c "$pick" chunk$c | tee -a gpurun_out/r03aw/chunk.txt
done
