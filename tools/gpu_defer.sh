#!/bin/bash
set -o pipefail
TAG=${1:-r4defer}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "deferred_text or train_step or first_touch or load_state_dict or kept_alive" > $OUT/pytest.log 2>&1; rc=$?; grep -E "deferred vs|passed|failed" $OUT/pytest.log | head -20; [ $rc -eq 0 ] || { tail -60 $OUT/pytest.log; exit $rc; }
bash tools/gpu_ab_env.sh $TAG 2 "CE_DEFER_TEXT_UPDATE=0 --" "CE_DEFER_TEXT_UPDATE=1 CE_ADAM_TEXT_GRID=32 --" "CE_DEFER_TEXT_UPDATE=1 CE_ADAM_TEXT_GRID=128 --" "CE_DEFER_TEXT_UPDATE=1 CE_ADAM_TEXT_GRID=512 --" "CE_DEFER_TEXT_UPDATE=1 CE_TXT_STREAM_PRIORITY=1 CE_ADAM_TEXT_GRID=128 --"
