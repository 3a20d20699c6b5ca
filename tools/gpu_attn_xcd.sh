#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-attnxcd}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -k "attention or attn or patch14 or vit_l14" > $OUT/pytest.log 2>&1; rc=$?; tail -2 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -50 $OUT/pytest.log; exit $rc; }
for r in 1 2; do for x in 1 0; do
  echo "CE_ATTN_XCD=$x $(CE_ATTN_XCD=$x python tools/diag/attn_long_time.py)"
done; done
for x in 1 0; do
  CE_ATTN_XCD=$x python bench.py --arch vit_l14_336 --no-cpu-baseline --no-roofline --steps 5 --warmup 2 2>$OUT/c5.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 bf16 CE_ATTN_XCD=$x', d['ms_per_step'])"
done
