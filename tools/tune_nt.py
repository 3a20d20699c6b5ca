#!/usr/bin/env python3
"""Sweep the NT GEMM tile variants (ce_gemm_nt_tune) over tower shapes; prints microseconds per variant.
Used to calibrate the tile-height cost model in csrc/gemm.hip:launch_nt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import ops, _lib as L

DEV = "cuda:0"
VARIANTS = [0, 3, 4, 5, 6, 7, 8, 32]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    torch.manual_seed(0)
    lib = L.lib()
    Ms = [int(a) for a in sys.argv[1:]] or [11137, 8000, 14000, 19712, 12800]
    nk = {"t": [(1536, 512), (512, 512), (2048, 512), (512, 2048), (512, 1536)],
          "v": [(2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)]}
    print("shape".ljust(26) + "".join(f"{v:>9d}" for v in VARIANTS) + "   best")
    for M in Ms:
        for tower in ("t", "v"):
            if tower == "v" and M != 12800:
                continue
            for N, K in nk[tower]:
                a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
                b = (torch.randn(N, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
                out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
                row = []
                for v in VARIANTS:
                    lib.ce_gemm_nt_tune(v)
                    row.append(timeit(lambda: ops.gemm_nt(a, b, L.EPI_BF16, out=out)))
                lib.ce_gemm_nt_tune(0)
                best = VARIANTS[1:][min(range(len(row) - 1), key=lambda i: row[i + 1])]
                print(f"M={M:6d} N={N:5d} K={K:5d}".ljust(26) + "".join(f"{t:9.1f}" for t in row) +
                      f"   {best} ({2*M*N*K/min(row[1:])/1e6:.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
