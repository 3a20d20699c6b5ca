#!/bin/bash
# kernel trace of the default two-stream bench -> idle accounting + one step's timeline
set -o pipefail
TAG=${1:-trace}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3 > $OUT/trace_bench.log 2>&1 || exit 1
cd $ROOT
python3 tools/trace_idle.py $OUT/trace $OUT/timeline_two_stream.txt > $OUT/trace_idle_two_stream.txt || exit 1
head -3 $OUT/trace_idle_two_stream.txt
rm -rf $OUT/trace
