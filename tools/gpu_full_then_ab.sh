#!/bin/bash
# full GPU suite, then the small-head A/B on the headline step
set -o pipefail
TAG=${1:-r4full}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -60 $OUT/pytest.log; exit $rc; }
bash tools/gpu_ab_env.sh $TAG 3 "CE_SMALL_HEAD=1 --" "CE_SMALL_HEAD=0 --"
