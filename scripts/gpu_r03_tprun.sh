set -o pipefail
OUT=gpurun_out/r03n; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_all.log)"
pick='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "value %.0f" % d["value"], "exclusive", {k: round(v,3) for k,v in r["exclusive_launch_ms"].items()})'
for C in 7 1; do for RL in 2 4 8 16; do
  BFIR_PAIR_RUN_FWD=$RL BFIR_PAIR_RUN_INV=$RL timeout -k 10 300 python bench.py --channels $C --blocks 65536 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>$OUT/err.log | python -c "$pick" C${C}_run$RL | tee -a $OUT/tprun.txt
done; done
