#!/bin/bash
# kernel trace of the config-4 step (B = 64, K = 5, alignment + region branch): idle accounting
set -o pipefail
TAG=${1:-tracec4}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3 > $OUT/trace_bench.log 2>&1 || { tail $OUT/trace_bench.log; exit 1; }
cd $ROOT
python3 tools/trace_idle.py $OUT/trace $OUT/timeline.txt > $OUT/trace_idle.txt || exit 1
head -14 $OUT/trace_idle.txt
tail -2 $OUT/trace_bench.log | cut -c1-300
rm -rf $OUT/trace
