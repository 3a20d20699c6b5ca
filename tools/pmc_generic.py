#!/usr/bin/env python3
"""Sum arbitrary rocprofv3 PMC counters per kernel: python3 tools/pmc_generic.py OUT_DIR [top_n] -> table on stdout.
(Collect with: rocprofv3 --pmc C1 C2 ... --kernel-trace -d OUT_DIR --output-format csv -- python3 bench.py ...)"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        a = acc[name]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), path)
        if key not in seen[name]:
            seen[name].add(key)
            a["_n"] += 1
            a["_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
names = sorted({k for a in acc.values() for k in a if not k.startswith("_")})
print("kernel".ljust(44), "n".rjust(5), "avg_us".rjust(8), *[n[-22:].rjust(22) for n in names])
for name, a in sorted(acc.items(), key=lambda kv: -kv[1]["_us"])[:top]:
    n = a["_n"]
    print(name[:44].ljust(44), str(int(n)).rjust(5), f"{a['_us'] / n:8.1f}", *[f"{a.get(k, 0.0) / n:22.4g}" for k in names])
