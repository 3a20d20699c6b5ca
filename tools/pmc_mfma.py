#!/usr/bin/env python3
"""MFMA utilisation per kernel from ONE rocprofv3 PMC pass (north star: "MFMA-utilisation counters reported against
chip peak") -> JSON under profiles/.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
            --kernel-trace -d OUT/mfma --output-format csv -- python3 bench.py --single-stream \
            --no-cpu-baseline --no-roofline --no-dense-compare --steps 3 --warmup 1
  python3 tools/pmc_mfma.py OUT/mfma > profiles/rNN_pmc_mfma_util.json

Per kernel (summed over launches, then divided):
  mfma_util       = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs): rocprofv3's own MfmaUtil formula
                    (counters.txt: reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum)/(reduce(GRBM_GUI_ACTIVE,max)*SIMD_NUM)) -- the share
                    of the chip's SIMD-cycles in which the matrix pipe was executing (MI355X_MICROARCH.md cycle constants:
                    the counter counts cycles, 32 per v_mfma_f32_32x32x16_bf16 / 16 per 16x16x32);
  mfma_busy_on_cu = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the same on the CUs the kernel occupied;
  mfma_tflops     = 512 FLOP x SQ_INSTS_VALU_MFMA_MOPS_BF16 / kernel time when that counter is present
                    (one MOP = 512 FLOP on the gfx94x/gfx950 definition), else null;
  eff_clock_ghz   = GRBM_GUI_ACTIVE / 8 XCDs / kernel time (the clock the chip held: 'DVFS give-back');
  frac_of_peak    = mfma_util x eff_clock / 2.4 GHz: busy share priced at the datasheet clock, i.e. against
                    the 2.5 PFLOP/s dense bf16 peak.
Counters a pass did not collect come out as null.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
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
    out = []
    for name, a in sorted(acc.items(), key=lambda kv: -kv[1]["_us"]):
        n, us = a["_n"], a["_us"]
        busy, cu = a.get("SQ_VALU_MFMA_BUSY_CYCLES"), a.get("SQ_BUSY_CU_CYCLES")
        mops, gui = a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"), a.get("GRBM_GUI_ACTIVE")
        on_cu = busy / (4.0 * cu) if busy is not None and cu else None
        frac = busy / (gui / 8.0 * 1024.0) if busy is not None and gui else None
        clock = gui / 8.0 / (us * 1e3) if gui and us else None
        out.append({"kernel": name, "launches": int(n), "avg_us_under_pmc": round(us / max(n, 1), 1),
                    "mfma_util": None if frac is None else round(frac, 4),
                    "mfma_busy_on_cu": None if on_cu is None else round(on_cu, 4),
                    "mfma_tflops": None if not mops else round(512.0 * mops / (us * 1e-6) / 1e12, 1),
                    "eff_clock_ghz": None if clock is None else round(clock, 3),
                    "frac_of_peak": None if frac is None or clock is None else round(frac * clock / 2.4, 4)})
    json.dump(out[:40], sys.stdout, indent=1)


if __name__ == "__main__":
    main()
