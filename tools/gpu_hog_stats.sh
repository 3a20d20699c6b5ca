#!/bin/bash
# per-kernel average durations: single-stream step with and without a one-wave hog on another stream (which kernels stretch?)
set -o pipefail
TAG=${1:-r4hogstats}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BARGS="--single-stream --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3"
rocprofv3 --kernel-trace --stats -d $OUT/p0 --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/b0.log 2>&1 || exit 1
CE_HOG_LDS=1024 CE_HOG_THREADS=64 rocprofv3 --kernel-trace --stats -d $OUT/p1 --output-format csv -- python3 $ROOT/bench.py $BARGS --cu-hog 1 > $OUT/b1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/p2 --output-format csv -- python3 $ROOT/bench.py $BARGS --cu-hog 1 > $OUT/b2.log 2>&1 || exit 1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, os, sys, re
out = sys.argv[1]
def load(d):
    f = glob.glob(os.path.join(out, d, "**", "*kernel_stats.csv"), recursive=True)[0]
    return {re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"]).split("(")[0][:50]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b, c = load("p0"), load("p1"), load("p2")
print(f"{'kernel':50s} {'calls':>6s} {'no hog us':>10s} {'1-wave hog':>10s} {'ratio':>6s} {'96KiB hog':>10s} {'ratio':>6s}")
for k, (n, us, tot) in sorted(a.items(), key=lambda kv: -kv[1][2])[:26]:
    if k in b and k in c:
        print(f"{k:50s} {n:6d} {us:10.1f} {b[k][1]:10.1f} {b[k][1] / us:6.2f} {c[k][1]:10.1f} {c[k][1] / us:6.2f}")
PY
rm -rf $OUT/p0 $OUT/p1 $OUT/p2
