#!/bin/bash
# round 4, first GPU pass: whole -m gpu suite (incl. the trained-like-statistics / saturation tests), then the bench lines:
# default, --device-lengths (A/B of the host-side caption lengths), config 4 and config 5 with their roofline objects
set -o pipefail
TAG=${1:-r4a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
grep -E "^\[|counters|stream16=" $OUT/pytest.log | grep -i "hostile\|tiny stream16\|vit_b32_b8\|counters" | head -20
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('host lengths  ', d['ms_per_step'])" || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare --device-lengths 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('device lengths', d['ms_per_step'])" || exit 1
done
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err || { tail -20 $OUT/bench_config4.err; exit 1; }
python bench.py --arch vit_l14_336 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_bf16.json 2> $OUT/bench_config5_bf16.err || { tail -20 $OUT/bench_config5_bf16.err; exit 1; }
python bench.py --arch vit_l14_336 --fp8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_fp8.json 2> $OUT/bench_config5_fp8.err || { tail -20 $OUT/bench_config5_fp8.err; exit 1; }
python - $OUT <<'PY'
import json, sys, os
for f in ("bench.json", "bench_config4.json", "bench_config5_bf16.json", "bench_config5_fp8.json"):
    d = json.load(open(os.path.join(sys.argv[1], f)))
    r = d["roofline"]
    print(f, d["ms_per_step"], "ms", d["value"], "pairs/s", "step_frac", r["step_frac"], "model", r["step_frac_model_live_text_rows"], "nominal", r["step_frac_nominal"], "issued TF", r["issued_tflop_per_step"])
    for c in r["classes"][:12]:
        print(f"   {c['kernel']:50s} n={c['launches_per_step']:6.1f} ms={c['ms_per_step']:.3f} us={c['avg_us']:8.1f} TF={c['tflops']:8.1f} GB/s={c['gbps']:8.1f}")
PY
