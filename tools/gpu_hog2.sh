#!/bin/bash
# CU budget under contention: step time for CE_GEMM_CUS x cu_hog (96 KiB / 256-thread hogs, the default)
set -o pipefail
TAG=${1:-r4hog6}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for cus in ${CUSS:-256 248 240 224}; do
  for k in ${KS:-0 8 16 32}; do
    ms=$(CE_GEMM_CUS=$cus python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare --cu-hog $k 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || { tail -5 $OUT/err.log; exit 1; }
    echo "CE_GEMM_CUS=$cus cu_hog=$k ms_per_step=$ms" | tee -a $OUT/hog.txt
  done
done
