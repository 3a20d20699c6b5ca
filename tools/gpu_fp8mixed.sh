#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-fp8mixed}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -s -k "fp8" > $OUT/pytest.log 2>&1; rc=$?; grep -E "fp8 two heights [0-9]|passed|failed" $OUT/pytest.log | head; [ $rc -eq 0 ] || { tail -50 $OUT/pytest.log; exit $rc; }
for rep in 1 2; do for m in 1 0; do
  CE_NT_MIXED=$m python bench.py --arch vit_l14_336 --fp8 --no-cpu-baseline --no-roofline --steps 5 --warmup 2 2>$OUT/c5.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 fp8 CE_NT_MIXED=$m', d['ms_per_step'])"
done; done
