#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3vsr4c4}; mkdir -p $OUT
export TMPDIR=/tmp
run() { python $1 --batch 64 --descriptions 5 --alignment --train-arg desc --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || { tail -5 $OUT/err.log; exit 1; }; }
for r in 1 2; do
  echo "config 4 round $r  round-3 tree $(run _r3/bench.py)  HEAD $(run bench.py)" | tee -a $OUT/ab.txt
done
