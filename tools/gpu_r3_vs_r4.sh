#!/bin/bash
# same-box comparison of the round-3 tree (git worktree _r3 at 1447c63, built in place) with HEAD: interleaved default bench runs
set -o pipefail
OUT=gpurun_out/${1:-r3vsr4}; mkdir -p $OUT
export TMPDIR=/tmp
run() { python $1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || { tail -5 $OUT/err.log; exit 1; }; }
for r in 1 2 3; do
  echo "round $r  round-3 tree $(run _r3/bench.py)  HEAD $(run bench.py)" | tee -a $OUT/ab.txt
done
