#!/usr/bin/env python3
"""Would sorting the captions by length pay?  Text-tower attention (8 heads, 256 captions of U[10,77] live tokens, packed) as one launch
sized for the longest caption, against two launches over the length-sorted batch (captions of <= 64 tokens with the 4-tile kernels, the
rest with the 6-tile ones).  Timing only."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from clip_event_amd import ops
dev = torch.device("cuda", 0)
B, H = 256, 8
rng = np.random.default_rng(0)
lens = rng.integers(10, 78, size=B)
def cu_of(l): return torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device=dev)
rows = int(lens.sum())
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(rows, 3 * H * 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dout = (torch.randn(rows, H * 64, generator=g) * 0.1).to(torch.bfloat16).to(dev)
bg = torch.zeros(3 * H * 64, device=dev)
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cu = cu_of(lens)
o, lse = ops.attention_fwd(qkv, B, 77, H, True, cu_seqlens=cu)
f1 = timeit(lambda: ops.attention_fwd(qkv, B, 77, H, True, cu_seqlens=cu))
b1 = timeit(lambda: ops.attention_bwd(qkv, o, dout, lse, B, 77, H, True, bias_grad=bg, cu_seqlens=cu))
print(f"one launch (Lmax 77): fwd {f1:.1f} us, bwd {b1:.1f} us")
for cut in (64, 32):
    sl = np.sort(lens)
    cus = cu_of(sl)
    ns = int((sl <= cut).sum())
    Ls = int(sl[ns - 1]) if ns else 0
    def fwd2():
        a = ops.attention_fwd(qkv, ns, cut, H, True, cu_seqlens=cus[:ns + 1])
        b = ops.attention_fwd(qkv, B - ns, 77, H, True, cu_seqlens=cus[ns:])
        return a, b
    (oa, la), (ob, lb) = fwd2()
    def bwd2():
        ops.attention_bwd(qkv, oa, dout, la, ns, cut, H, True, bias_grad=bg, cu_seqlens=cus[:ns + 1])
        ops.attention_bwd(qkv, ob, dout, lb, B - ns, 77, H, True, bias_grad=bg, cu_seqlens=cus[ns:])
    print(f"sorted, {ns} captions <= {cut} tokens + {B - ns} longer: fwd {timeit(fwd2):.1f} us, bwd {timeit(bwd2):.1f} us")
