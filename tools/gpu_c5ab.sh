#!/bin/bash
set -o pipefail
TAG=${1:-r4c5ab}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in ${CFGS:-"1 1" "2 1"}; do
  set -- $cfg
  CE_ATTN_NS=$1 CE_ATTN_NK=$2 timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "attention or attn" > $OUT/pytest_attn_$1$2.log 2>&1 || { tail -30 $OUT/pytest_attn_$1$2.log; exit 1; }
  CE_ATTN_NS=$1 CE_ATTN_NK=$2 python bench.py --arch vit_l14_336 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_$1$2.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
  python - $OUT/bench_$1$2.json "NS=$1 NK=$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
cl = {c["kernel"]: c for c in r["classes"]}
print(sys.argv[2], d["ms_per_step"], "ms | attn_bwd", round(cl["attn_bwd"]["ms_per_step"], 3), "ms", round(cl["attn_bwd"]["tflops"]), "TF/s | attn_fwd", round(cl["attn_fwd"]["ms_per_step"], 3), "ms", round(cl["attn_fwd"]["tflops"]), "TF/s")
PY
done
