#!/bin/bash
# the full GPU suite, log under gpurun_out/<tag>/pytest.log
set -o pipefail
TAG=${1:-suite}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -80 $OUT/pytest.log; exit $rc; }
grep -h "\[sharded\]" $OUT/pytest.log | head || true
