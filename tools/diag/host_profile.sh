#!/bin/bash
# cProfile of the host side of bench.py's step (run-ahead unlimited): top cumulative entries
set -o pipefail
OUT=gpurun_out/${1:-hostprof}; mkdir -p $OUT
shift
export TMPDIR=/tmp
CE_STEPS_AHEAD=0 python -m cProfile -o $OUT/prof.bin bench.py "$@" --no-cpu-baseline --no-roofline --no-dense-compare --steps 30 --warmup 3 > $OUT/b.json 2> $OUT/b.err || { tail $OUT/b.err; exit 1; }
python - $OUT/prof.bin <<'PY'
import pstats, sys
p = pstats.Stats(sys.argv[1])
p.sort_stats("cumulative").print_stats(45)
PY
