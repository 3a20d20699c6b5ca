#!/bin/bash
# same-box A/B of bench.py under different environments / flags: tools/gpu_ab_env.sh TAG ROUNDS "ENV1 -- flags1" "ENV2 -- flags2" ...
set -o pipefail
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
  i=0
  for spec in "$@"; do
    i=$((i+1))
    envs="${spec%%--*}"; flags="${spec#*--}"
    [ "$spec" == "$envs" ] && flags=""
    ms=$(env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare $flags 2>$OUT/err_$i.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || { tail -5 $OUT/err_$i.log; exit 1; }
    echo "round $r  [$envs|$flags]  $ms" | tee -a $OUT/ab.txt
  done
done
