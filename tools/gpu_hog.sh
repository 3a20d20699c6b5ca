#!/bin/bash
# DESIGN 5 / VERDICT r3 item 6: step time with k CUs held by a "CU hog" (what an RCCL ring's channel kernels take) for
# CE_NT_PGRID 256 / 512 / 1024 and GPU_MAX_HW_QUEUES 4 (ROCm default) / 8, plus the CE_FORCE_COLLECTIVES=1 line
set -o pipefail
TAG=${1:-r4hog}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for hq in ${HQS:-4 8}; do
for pg in ${PGS:-256 1024}; do
  for k in ${KS:-0 8 32}; do
    ms=$(GPU_MAX_HW_QUEUES=$hq CE_NT_PGRID=$pg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare --cu-hog $k 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || { tail -5 $OUT/err.log; exit 1; }
    echo "GPU_MAX_HW_QUEUES=$hq CE_NT_PGRID=$pg cu_hog=$k ms_per_step=$ms" | tee -a $OUT/hog.txt
  done
done
done
for hq in 4 8; do
GPU_MAX_HW_QUEUES=$hq CE_FORCE_COLLECTIVES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare > $OUT/bench_force_collectives_$hq.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python -c "
import json
a=json.load(open('$OUT/bench_force_collectives_$hq.json'))
print('GPU_MAX_HW_QUEUES=$hq CE_FORCE_COLLECTIVES=1', a['ms_per_step'], 'rccl_ranks', a['rccl_ranks'])" | tee -a $OUT/hog.txt
done
