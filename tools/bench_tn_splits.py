#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import ops
from tools.bench_ops import timeit
DEV = "cuda:0"
for name, M, N, K in [("v.fc", 12800, 3072, 768), ("v.qkv", 12800, 2304, 768), ("v.out", 12800, 768, 768), ("t.fc", 19712, 2048, 512), ("t.out", 19712, 512, 512)]:
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    g = torch.zeros(N, K, device=DEV)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    row = []
    for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
        t = timeit(lambda: ops.gemm_tn(dy, a, g, splits=sp))
        row.append(f"s{sp}:{2*M*N*K/t/1e12:5.0f}")
    print(f"{name:6s} tiles={tiles:4d} " + " ".join(row), flush=True)
