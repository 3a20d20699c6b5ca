#!/usr/bin/env python3
"""Times the three long-sequence attention kernels at the ViT-L/14@336 geometry (B = 32, 16 heads, L = 577) in a loop of their
own: forward, then the backward (dQ kernel + dK/dV kernel), per-call averages from HIP events."""
import sys, torch
sys.path.insert(0, ".")
from clip_event_amd import ops
B, H, L = 32, 16, 577
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
qkv = (torch.randn(B * L, 3 * H * 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dout = (torch.randn(B * L, H * 64, generator=g) * 0.1).to(torch.bfloat16).to(dev)
bg = torch.zeros(3 * H * 64, device=dev)
o, lse = ops.attention_fwd(qkv, B, L, H, False)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tf = timeit(lambda: ops.attention_fwd(qkv, B, L, H, False))
tb = timeit(lambda: ops.attention_bwd(qkv, o, dout, lse, B, L, H, False, bias_grad=bg))
print(f"fwd {tf:.1f} us  bwd (dq + dkv) {tb:.1f} us")
