#!/bin/bash
set -o pipefail
TAG=${1:-r4tiles}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "adam or train_step or load_state_dict or first_touch or deferred or checkpoint" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -60 $OUT/pytest.log; exit $rc; }
bash tools/gpu_ab_env.sh $TAG 3 "CE_ADAM_TILES=1 --" "CE_ADAM_TILES=0 --"
