#!/bin/bash
# diagnostic builds of attention.hip (wrong results, timing only): where the long dK/dV kernel's time goes
set -o pipefail
OUT=gpurun_out/${1:-attn_ablate}; mkdir -p $OUT
export TMPDIR=/tmp
for flags in "" "-DCE_ABL_NO_DELTA" "-DCE_ABL_NO_EXP" "-DCE_ABL_NO_SECOND" "-DCE_ABL_NO_DELTA -DCE_ABL_NO_EXP -DCE_ABL_NO_SECOND"; do
  touch clip_event_amd/csrc/attention.hip
  CE_EXTRA_FLAGS="$flags" python -m clip_event_amd.build > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
  echo "[$flags] $(python tools/diag/attn_long_time.py 2>>$OUT/err.log)" | tee -a $OUT/ablate.txt
done
touch clip_event_amd/csrc/attention.hip
python -m clip_event_amd.build > $OUT/build.log 2>&1
