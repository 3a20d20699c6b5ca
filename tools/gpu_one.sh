#!/bin/bash
# tools/gpu_one.sh TAG "pytest args": one pytest invocation on the GPU box, log under gpurun_out/TAG
set -o pipefail
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest $2 > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; [ $rc -eq 0 ] || { tail -60 $OUT/pytest.log; exit $rc; }
