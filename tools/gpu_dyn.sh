#!/bin/bash
set -o pipefail
TAG=${1:-r4dyn}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 240 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "dynamic_tile" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -40 $OUT/pytest.log; exit $rc; }
run() { timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare "$@" 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" || { tail -5 $OUT/err.log; exit 1; }; }
for r in 1 2; do
echo "static : $(CE_NT_DYNAMIC=0 run) | hog 8x96KiB: $(CE_NT_DYNAMIC=0 run --cu-hog 8) | hog 1 wave: $(CE_NT_DYNAMIC=0 CE_HOG_LDS=1024 CE_HOG_THREADS=64 run --cu-hog 1)" | tee -a $OUT/dyn.txt
echo "dynamic: $(CE_NT_DYNAMIC=1 run) | hog 8x96KiB: $(CE_NT_DYNAMIC=1 run --cu-hog 8) | hog 1 wave: $(CE_NT_DYNAMIC=1 CE_HOG_LDS=1024 CE_HOG_THREADS=64 run --cu-hog 1)" | tee -a $OUT/dyn.txt
done
