#!/bin/bash
# config 5 (ViT-L/14@336) check: long-attention parity tests, then the bench class table
set -o pipefail
TAG=${1:-r4c5}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "attention or attn" > $OUT/pytest_attn.log 2>&1; rc=$?; tail -3 $OUT/pytest_attn.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "patch14 or vit_l14 or 577 or long" > $OUT/pytest_model.log 2>&1; rc=$?; tail -3 $OUT/pytest_model.log; [ $rc -eq 0 ] || exit $rc
python bench.py --arch vit_l14_336 --steps 5 --warmup 2 --no-cpu-baseline ${C5FLAGS} > $OUT/bench_config5.json 2> $OUT/bench_config5.err || { tail -20 $OUT/bench_config5.err; exit 1; }
python - $OUT <<'PY'
import json, sys, os
d = json.load(open(os.path.join(sys.argv[1], "bench_config5.json")))
r = d["roofline"]
print(d["ms_per_step"], "ms", d["value"], "pairs/s", "step_frac", r["step_frac"], "nominal", r["step_frac_nominal"], d.get("power_sample"))
for c in r["classes"][:10]:
    print(f"   {c['kernel']:50s} n={c['launches_per_step']:6.1f} ms={c['ms_per_step']:.3f} us={c['avg_us']:8.1f} TF={c['tflops']:8.1f} GB/s={c['gbps']:8.1f}")
PY
