#!/bin/bash
# builds of optim.hip with diagnostic flags, timing FusedAdam.step() in both modes
set -o pipefail
OUT=gpurun_out/${1:-adamvar}; mkdir -p $OUT
export TMPDIR=/tmp
shift
for flags in "$@"; do
  touch clip_event_amd/csrc/optim.hip
  CE_EXTRA_FLAGS="$flags" python -m clip_event_amd.build > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
  echo "[$flags] $(CE_ADAM_TILES=1 python tools/diag/adam_time.py 2>>$OUT/err.log) | $(CE_ADAM_TILES=0 python tools/diag/adam_time.py 2>>$OUT/err.log)" | tee -a $OUT/variants.txt
done
touch clip_event_amd/csrc/optim.hip; python -m clip_event_amd.build > $OUT/build.log 2>&1
