#!/bin/bash
set -o pipefail
TAG=${1:-r4mixed}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -s -k "two_tile_heights or dynamic_tile_list or persistent or epilogues" > $OUT/pytest.log 2>&1; rc=$?; grep -E "^two heights|passed|failed" $OUT/pytest.log | sort | uniq | head -20; [ $rc -eq 0 ] || { tail -60 $OUT/pytest.log; exit $rc; }
bash tools/gpu_ab_env.sh $TAG 3 "CE_NT_MIXED=1 --" "CE_NT_MIXED=0 --"
