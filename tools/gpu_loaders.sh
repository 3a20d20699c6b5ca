#!/bin/bash
# diagnostic: the loader-wave NT kernels with 2 instead of 4 loader waves (10 waves per workgroup: two SIMDs of every CU keep 176
# VGPRs free) -- step time, and the penalty of a one-wave hog beside it
set -o pipefail
TAG=${1:-r4ld}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare "$@" 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" || { tail -5 $OUT/err.log; exit 1; }; }
echo "4 loaders: $(run) | hog1 tiny: $(CE_HOG_LDS=1024 CE_HOG_THREADS=64 run --cu-hog 1) | single-stream: $(run --single-stream)" | tee -a $OUT/ld.txt
CE_EXTRA_FLAGS="-DCE_N4_LOADERS=2" python -m clip_event_amd.build --force > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm_nt" > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
echo "2 loaders: $(run) | hog1 tiny: $(CE_HOG_LDS=1024 CE_HOG_THREADS=64 run --cu-hog 1) | single-stream: $(run --single-stream)" | tee -a $OUT/ld.txt
echo "2 loaders again: $(run)" | tee -a $OUT/ld.txt
