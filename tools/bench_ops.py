#!/usr/bin/env python3
"""Micro-benchmarks of the HIP ops at the ViT-B/32 (B=256) shapes; prints TFLOP/s / GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import ops, _lib as L

DEV = "cuda:0"


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
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    torch.manual_seed(0)
    shapes = [("v.qkv", 12800, 2304, 768), ("v.out", 12800, 768, 768), ("v.fc", 12800, 3072, 768),
              ("v.proj", 12800, 768, 3072), ("t.qkv", 19712, 1536, 512), ("t.out", 19712, 512, 512),
              ("t.fc", 19712, 2048, 512), ("t.proj", 19712, 512, 2048), ("sq4096", 4096, 4096, 4096)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        b = (torch.randn(N, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt(a, b, L.EPI_BF16, out=out))
        print(f"nt  {name:8s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        # wgrad: out[N,K] += dY[M,N]^T X[M,K]
        dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        g = torch.zeros(N, K, device=DEV)
        t = timeit(lambda: ops.gemm_tn(dy, a, g))
        print(f"tn  {name:8s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        from ctypes import c_int, c_long
        from clip_event_amd._lib import check, lib, ptr, stream
        bg = torch.zeros(N, device=DEV)
        t = timeit(lambda: check(lib().ce_gemm_tn_bias(ptr(dy), c_long(N), ptr(a), c_long(K), c_int(M), c_int(N), c_int(K), ptr(g), c_long(K), ptr(bg), c_int(0), stream()), "tnb"))
        print(f"tnb {name:8s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
    for M, D in [(12800, 768), (19712, 512)]:
        x = torch.randn(M, D, device=DEV)
        w = torch.ones(D, device=DEV); b = torch.zeros(D, device=DEV)
        t = timeit(lambda: ops.layernorm_fwd(x, w, b))
        print(f"ln_fwd M={M} D={D}: {t*1e6:8.1f} us  {M*D*6/t/1e9:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
