#!/usr/bin/env python3
"""Standalone A/B of forced NT tile variants (ce_gemm_nt_tune codes), warm (same operands back to back) and cold (a 1 GiB
fill between launches, timed separately and subtracted); checks each variant against variant 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clip_event_amd import ops, _lib as L

DEV = "cuda:0"
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,104,5,160,161,32").split(",")]
EPI = int(os.environ.get("EPI", "0"))          # 0 BF16, 2 BIAS_BF16, 4 BIAS_RESID_F32, 5 BIAS_GELU, 6 GELUGRAD_BF16


def timeit(fn, pre=None, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


def main():
    torch.manual_seed(0)
    lib = L.lib()
    junk = torch.empty(1 << 28, device=DEV, dtype=torch.float32)
    shapes = [(12800, 768, 768), (12800, 768, 3072), (12800, 768, 2304), (12800, 2304, 768), (12800, 3072, 768),
              (11137, 512, 512), (11137, 512, 2048), (11137, 2048, 512)]
    print("shape".ljust(28) + "".join(f"{v:>14d}" for v in VARIANTS) + "   (us warm/cold)")
    for M, N, K in shapes:
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        b = (torch.randn(N, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
        ref = None
        row = []
        bias = torch.randn(N, device=DEV)
        resid = torch.randn(M, N, device=DEV) if EPI == 4 else None
        aux = torch.randn(M, N, device=DEV).to(torch.bfloat16) if EPI == 6 else None
        colsum = torch.zeros(N, device=DEV) if EPI == 6 else None
        for v in VARIANTS:
            lib.ce_gemm_nt_tune(v)
            out = torch.empty(M, N, device=DEV, dtype=torch.float32 if EPI == 4 else torch.bfloat16)
            out2 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16) if EPI == 5 else colsum

            def run():
                return ops.gemm_nt(a, b, EPI, bias=bias if EPI in (2, 4, 5) else None, resid=resid, aux=aux, out=out, out2=out2)

            run()
            if ref is None:
                ref = out.float().clone()
            else:
                err = (out.float() - ref).abs().max().item()
                assert err < 2e-2 * ref.abs().max().item(), (v, M, N, K, err)
            w = timeit(run)
            c = timeit(run, pre=lambda: junk.fill_(1.0))
            row.append((w, c))
        lib.ce_gemm_nt_tune(0)
        print(f"M={M:6d} N={N:5d} K={K:5d}".ljust(28) + "".join(f"{w:7.1f}/{c:6.1f}" for w, c in row), flush=True)


if __name__ == "__main__":
    main()
