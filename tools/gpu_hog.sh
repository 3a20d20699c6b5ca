#!/bin/bash
# DESIGN 5 / VERDICT r3 item 6: step time with k CUs held by a "CU hog" (what an RCCL ring's channel kernels take) for
# CE_NT_PGRID 256 / 512 / 1024, plus the CE_FORCE_COLLECTIVES=1 line (every collective through RCCL with one rank)
set -o pipefail
TAG=${1:-r4hog}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for pg in 256 512 1024; do
  for k in 0 8 16 32; do
    ms=$(CE_NT_PGRID=$pg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare --cu-hog $k 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || { tail -5 $OUT/err.log; exit 1; }
    echo "CE_NT_PGRID=$pg cu_hog=$k ms_per_step=$ms" | tee -a $OUT/hog.txt
  done
done
CE_FORCE_COLLECTIVES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare > $OUT/bench_force_collectives.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare > $OUT/bench_default.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python -c "
import json
a=json.load(open('$OUT/bench_force_collectives.json')); b=json.load(open('$OUT/bench_default.json'))
print('CE_FORCE_COLLECTIVES=1', a['ms_per_step'], 'rccl_ranks', a['rccl_ranks'], '| default', b['ms_per_step'])" | tee -a $OUT/hog.txt
