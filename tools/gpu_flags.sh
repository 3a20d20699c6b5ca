#!/bin/bash
# same-box A/B of diagnostic BUILDS: tools/gpu_flags.sh TAG ROUNDS "flags1" "flags2" ... (each: CE_EXTRA_FLAGS for a forced rebuild of gemm.hip)
set -o pipefail
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" || { tail -5 $OUT/err.log; exit 1; }; }
for flags in "$@"; do
  touch clip_event_amd/csrc/gemm.hip
  CE_EXTRA_FLAGS="$flags" python -m clip_event_amd.build > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
  python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm_nt" > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
  line="[$flags]"
  for r in $(seq 1 $ROUNDS); do line="$line $(run)"; done
  echo "$line" | tee -a $OUT/flags.txt
done
