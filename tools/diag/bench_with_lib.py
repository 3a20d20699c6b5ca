#!/usr/bin/env python3
"""Run bench.py against an ablation build of the library (tools/diag/tn3_ablate.sh): timing only, results are wrong by
construction.      CE_DIAG_LIB=/tmp/libce_diag_1.so python tools/diag/bench_with_lib.py --steps 15 --warmup 4 ..."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import clip_event_amd._lib as L

L.LIB_PATH = os.environ["CE_DIAG_LIB"]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
