#!/usr/bin/env python3
"""Attention backward at the step's shapes with and without the in_proj bias-gradient output (its column sums end in
192 float atomics per (sample, head) workgroup onto 2,304 addresses): what the same-address atomics cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clip_event_amd import ops

DEV = "cuda:0"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, B, L, H, causal in (("image", 256, 50, 12, False), ("text dense", 256, 77, 8, True)):
    D = H * 64
    qkv = torch.randn(B * L, 3 * D, device=DEV).to(torch.bfloat16)
    o, lse = ops.attention_fwd(qkv, B, L, H, causal)
    do = torch.randn_like(o)
    bg = torch.zeros(3 * D, device=DEV)
    t1 = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, L, H, causal, bias_grad=bg))
    t0 = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, L, H, causal, bias_grad=None))
    tf = timeit(lambda: ops.attention_fwd(qkv, B, L, H, causal))
    print(f"{name}: backward {t1:.1f} us with bias gradient, {t0:.1f} us without; forward {tf:.1f} us")
