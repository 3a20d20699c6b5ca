#!/bin/bash
set -o pipefail
TAG=${1:-r4c5mixed}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for m in 1 0; do
  CE_NT_MIXED=$m python bench.py --arch vit_l14_336 --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $OUT/c5_$m.json 2> $OUT/c5_$m.err || { tail $OUT/c5_$m.err; exit 1; }
  python -c "import json; d=json.load(open('$OUT/c5_$m.json')); print('config5 bf16 CE_NT_MIXED=$m', d['ms_per_step'])"
  CE_NT_MIXED=$m python bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $OUT/c4_$m.json 2> $OUT/c4_$m.err || { tail $OUT/c4_$m.err; exit 1; }
  python -c "import json; d=json.load(open('$OUT/c4_$m.json')); print('config4 CE_NT_MIXED=$m', d['ms_per_step'])"
done
done
