#!/bin/bash
# One GPU-box session: tests, bench, config-4 bench, counters.  Usage: tools/gpu_round.sh TAG
set -o pipefail
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
cat $OUT/bench.json | cut -c1-600
