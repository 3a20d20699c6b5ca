#!/bin/bash
set -o pipefail
TAG=${1:-r4headtrace}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/p --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3 > $OUT/b.log 2>&1 || exit 1
cd $ROOT
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
grep -E "hs_|l2norm|xent|sgemm|dot_kernel" $f | cut -c1-200
python3 tools/trace_idle.py $OUT/p $OUT/timeline.txt | head -3
grep -n "hs_norm" -B3 -A12 $OUT/timeline.txt | head -40
rm -rf $OUT/p
