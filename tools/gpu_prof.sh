#!/bin/bash
# rocprofv3 kernel stats of the single-stream bench -> gpurun_out/$1/stats.csv (+ the run's own JSON)
TAG=$1
mkdir -p gpurun_out/$TAG
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG/prof --output-format csv -- python3 $ROOT/bench.py --single-stream --no-cpu-baseline --no-dense-compare --steps 10 --warmup 3 > $ROOT/gpurun_out/$TAG/bench_single_under_rocprof.json 2> $ROOT/gpurun_out/$TAG/prof.err || exit 1
cd $ROOT
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$TAG/stats.csv
rm -rf gpurun_out/$TAG/prof
python3 - gpurun_out/$TAG/stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 13 + 2 + 2   # warmup + timed + instrumented (2 warm + 2 profiled)
print(f"total kernel time {tot/1e6:.1f} ms over ~{steps} steps = {tot/1e6/steps:.2f} ms/step")
for r in rows[:26]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
