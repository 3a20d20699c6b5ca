#!/bin/bash
# kernel timeline of one step with a 1-CU hog on a fifth stream: where do the extra ~3 ms come from?
set -o pipefail
TAG=${1:-r4hogtrace}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3 --cu-hog 1 > $OUT/trace_bench.log 2>&1 || { tail -5 $OUT/trace_bench.log; exit 1; }
cd $ROOT
python3 tools/trace_idle.py $OUT/trace $OUT/timeline_hog.txt | tee $OUT/trace_idle_hog.txt
rm -rf $OUT/trace
