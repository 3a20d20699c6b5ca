#!/usr/bin/env python3
"""Grouped weight-gradient launch (ce_gemm_tn_grouped) at the step's shapes: one residual block of the image tower
(M = 12800, d = 768) and of the packed text tower (M = 11137, d = 512).  CE_GEMM_TN=2|3 picks the kernel (read once per
process): run once per value to compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from ctypes import c_int, c_long, c_void_p
import torch
import clip_event_amd._lib as _L
if os.environ.get("CE_DIAG_LIB"):          # tools/diag/tn3_ablate.sh: an ablation build of the library
    _L.LIB_PATH = os.environ["CE_DIAG_LIB"]
from clip_event_amd._lib import check, lib, stream

DEV = "cuda:0"


def group(M, d):
    Nn = [d, 4 * d, d, 3 * d]
    Kk = [4 * d, d, d, d]
    P = [torch.randn(M, n, device=DEV).to(torch.bfloat16) for n in Nn]
    Q = [(torch.randn(M, k, device=DEV) * M ** -0.5).to(torch.bfloat16) for k in Kk]
    out = [torch.zeros(n, k, device=DEV) for n, k in zip(Nn, Kk)]
    arr = lambda ts: (c_void_p * 4)(*[t.data_ptr() for t in ts])
    longs = lambda v: (c_long * 4)(*v)
    ints = lambda v: (c_int * 4)(*v)
    args = (c_int(4), arr(P), longs(Nn), arr(Q), longs(Kk), c_int(M), ints(Nn), ints(Kk), arr(out), longs(Kk), c_int(0))
    flops = sum(2.0 * M * n * k for n, k in zip(Nn, Kk))
    return args, flops, (P, Q, out)


def main():
    cl = lib()
    for name, M, d in (("image block", 12800, 768), ("text block (packed)", 11137, 512), ("text block (dense)", 19712, 512),
                       ("ViT-L/14 block", 18464, 1024)):
        args, flops, keep = group(M, d)
        fn = lambda: check(cl.ce_gemm_tn_grouped(*args, stream()), "tn")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"CE_GEMM_TN={os.environ.get('CE_GEMM_TN', 'default')} CE_TN3_SPLITS={os.environ.get('CE_TN3_SPLITS', '-')} "
              f"{name:22s} M={M} d={d}: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
