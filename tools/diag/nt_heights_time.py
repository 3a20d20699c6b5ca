#!/usr/bin/env python3
"""Per-launch time of the persistent NT GEMM at the towers' multi-round shapes for: the launcher's own choice (two tile heights
where it finds a better split), and the forced single heights 160 / 128 / 96 rows (ce_gemm_nt_tune 165 / 164 / 163).
Interleaved rounds of 60 launches per variant, so that clock state is shared."""
import sys, torch
from ctypes import byref, c_int
sys.path.insert(0, ".")
from clip_event_amd import ops, _lib as L
dev = torch.device("cuda", 0)
lib = L.lib()
shapes = [(12800, 3072, 768, "img c_fc"), (12800, 2304, 768, "img qkv"), (11136, 2048, 512, "txt c_fc"), (11136, 1536, 512, "txt qkv"),
          (18464, 4096, 1024, "vit-l c_fc"), (18464, 3072, 1024, "vit-l qkv")]
for M, N, K, name in shapes:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    B = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = {}
    for rnd in range(3):
        for var in (0, 165, 164, 163):
            lib.ce_gemm_nt_tune(var)
            for _ in range(5):
                ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=bias)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=bias)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(var, []).append(e0.elapsed_time(e1) / 60 * 1e3)
            if var == 0 and rnd == 0:
                t, s = c_int(), c_int(); lib.ce_gemm_nt_last_plan(byref(t), byref(s)); plan = (t.value, s.value)
    lib.ce_gemm_nt_tune(0)
    fl = 2.0 * M * N * K
    print(f"{name:11s} {M}x{N}x{K} plan {plan}: " + "  ".join(f"{'auto' if v == 0 else v}: {min(ts):6.1f} us ({fl / min(ts) / 1e6:5.0f} TF/s)" for v, ts in res.items()))
