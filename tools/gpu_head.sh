#!/bin/bash
set -o pipefail
TAG=${1:-r4head}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py tests/test_full_size_gpu.py -m gpu -x -q -s -k "small_head" > $OUT/pytest.log 2>&1; rc=$?; grep -E "loss_i|small head|rel-l2|passed|failed" $OUT/pytest.log | head -20; [ $rc -eq 0 ] || { tail -40 $OUT/pytest.log; exit $rc; }
bash tools/gpu_ab_env.sh $TAG 3 "CE_SMALL_HEAD=1 --" "CE_SMALL_HEAD=0 --"
