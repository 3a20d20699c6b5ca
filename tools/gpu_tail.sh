#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-tail}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_rccl_single_rank_gpu.py tests/test_ddp_gpu.py tests/test_stream16_hostile_gpu.py tests/test_tokenizer.py tests/test_preprocess.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -80 $OUT/pytest.log; exit $rc; }
grep -h "\[sharded\]" $OUT/pytest.log | head -12
