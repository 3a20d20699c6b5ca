#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-attn}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -k "attention or attn or patch14 or vit_l14" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -50 $OUT/pytest.log; exit $rc; }
python tools/diag/attn_long_time.py
python bench.py --arch vit_l14_336 --no-cpu-baseline --no-roofline --steps 5 --warmup 2 2>$OUT/c5.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 bf16', d['ms_per_step'])"
