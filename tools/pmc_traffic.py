#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each collected in its own
run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -> JSON under profiles/.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT/fetch --output-format csv -- python3 bench.py --single-stream \
            --no-cpu-baseline --no-roofline --steps 3 --warmup 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT/write --output-format csv -- python3 bench.py ... (same)
  python3 tools/pmc_traffic.py OUT/fetch OUT/write > profiles/rNN_pmc_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced
reads (128-B requests tallied at 64 B; guide section 'HBM traffic'), so `fetch_corrected_x2_MB` doubles it; the
truth for a kernel that mixes wide and narrow reads lies between the two columns.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(dirname, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = []
    for name, (n, kib, us) in sorted(fetch.items(), key=lambda kv: -kv[1][2]):
        w = write.get(name, [0, 0.0, 0.0])
        mb = kib * 1024 / 1e6 / n
        out.append({"kernel": name, "launches": n, "FETCH_SIZE_MB_per_launch": round(mb, 2),
                    "fetch_corrected_x2_MB": round(2 * mb, 2),
                    "WRITE_SIZE_MB_per_launch": round(w[1] * 1024 / 1e6 / max(w[0], 1), 2),
                    "avg_us_under_pmc": round(us / n, 1)})
    json.dump(out[:40], sys.stdout, indent=1)


if __name__ == "__main__":
    main()
