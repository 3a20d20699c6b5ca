#!/bin/bash
# same-box A/B of diagnostic BUILDS of one source file: tools/gpu_flags_file.sh TAG ROUNDS FILE "pytest -k expr" "flags1" "flags2" ...
# (rounds interleaved: every round rebuilds FILE with each flag set in turn)
set -o pipefail
TAG=$1; ROUNDS=$2; FILE=$3; KEXPR=$4; shift 4
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare 2>$OUT/err.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" || { tail -5 $OUT/err.log; exit 1; }; }
for r in $(seq 1 $ROUNDS); do
  for flags in "$@"; do
    touch clip_event_amd/csrc/$FILE
    CE_EXTRA_FLAGS="$flags" python -m clip_event_amd.build > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
    if [ $r -eq 1 ]; then python -m pytest tests/test_hip_ops.py tests/test_model_gpu.py -m gpu -x -q -k "$KEXPR" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }; fi
    echo "round $r [$flags] $(run)" | tee -a $OUT/ab.txt
  done
done
touch clip_event_amd/csrc/$FILE; python -m clip_event_amd.build > $OUT/build.log 2>&1
