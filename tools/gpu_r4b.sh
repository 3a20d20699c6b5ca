#!/bin/bash
# round 4, second GPU pass: the new tests + the config-4 parity gates on the merged (one pass per tower) step, then bench lines
set -o pipefail
TAG=${1:-r4b}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_stream16_hostile_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
grep -E "stream16=|counters|config4|combined" $OUT/pytest.log | head -30
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('host lengths  ', d['ms_per_step'])" || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare --device-lengths 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('device lengths', d['ms_per_step'])" || exit 1
done
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
CE_MERGE_PASSES=0 python bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config4_unmerged.json 2> $OUT/bench_config4_unmerged.err || { tail -20 $OUT/bench_config4_unmerged.err; exit 1; }
python bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err || { tail -20 $OUT/bench_config4.err; exit 1; }
python bench.py --arch vit_l14_336 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_bf16.json 2> $OUT/bench_config5_bf16.err || { tail -20 $OUT/bench_config5_bf16.err; exit 1; }
python bench.py --arch vit_l14_336 --fp8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_fp8.json 2> $OUT/bench_config5_fp8.err || { tail -20 $OUT/bench_config5_fp8.err; exit 1; }
python - $OUT <<'PY'
import json, sys, os
for f in ("bench.json", "bench_config4_unmerged.json", "bench_config4.json", "bench_config5_bf16.json", "bench_config5_fp8.json"):
    d = json.load(open(os.path.join(sys.argv[1], f)))
    r = d["roofline"]
    print(f, d["ms_per_step"], "ms", d["value"], "pairs/s", "step_frac", r["step_frac"], "model", r["step_frac_model_live_text_rows"], "nominal", r["step_frac_nominal"], "issued TF", r["issued_tflop_per_step"])
    for c in r["classes"][:12]:
        print(f"   {c['kernel']:50s} n={c['launches_per_step']:6.1f} ms={c['ms_per_step']:.3f} us={c['avg_us']:8.1f} TF={c['tflops']:8.1f} GB/s={c['gbps']:8.1f}")
PY
