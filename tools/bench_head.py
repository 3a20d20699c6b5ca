#!/usr/bin/env python3
"""Contrastive head at BASELINE config 3's per-rank shape (W = 8, B = 512, K = 5: 512 image rows against 20,480 gathered
text columns; 512 positive text rows against 4,096 gathered image columns), forward + backward: the fused kernels
(no logits matrix) against the logits + cross-entropy path.  Features are synthetic gathered matrices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd.functional import InfoNCEFn, logits_from_features
from clip_event_amd.losses import cross_entropy

DEV = "cuda:0"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    E = 512
    for B, K, W in ((512, 5, 8), (256, 1, 8), (256, 1, 1)):
        N = B * W
        fi = torch.randn(B, E, device=DEV, requires_grad=True)
        ft = torch.randn(B * K, E, device=DEV, requires_grad=True)
        fi_all = torch.randn(N, E, device=DEV, requires_grad=True)
        ft_all = torch.randn(N * K, E, device=DEV, requires_grad=True)
        ls = torch.tensor(2.659, device=DEV, requires_grad=True)
        yi = torch.arange(B, device=DEV) * K
        yt = torch.arange(B, device=DEV).repeat_interleave(K)
        ip = torch.arange(0, B * K, K, device=DEV)

        def fused():
            l = InfoNCEFn.apply(fi, ft_all, ls, yi, None) + InfoNCEFn.apply(ft, fi_all, ls, yt, ip)
            l.backward()

        def unfused():
            lpi, _ = logits_from_features(fi, ft_all, ls, True, want="image")
            _, lpt = logits_from_features(fi_all, ft, ls, True, want="text")
            l = cross_entropy(lpi, yi) + cross_entropy(lpt, yt, ip)
            l.backward()

        tf, tu = timeit(fused), timeit(unfused)
        flops = 3 * 2.0 * E * (B * N * K + B * N)          # fwd + two backward products, both directions (the fused path's work)
        print(f"B={B} K={K} W={W}: fused {tf:.3f} ms ({flops / tf / 1e9:.1f} TF/s on 6*nq*nk*E), logits+CE path {tu:.3f} ms "
              f"(logits_per_image {B * N * K * 4 / 1e6:.0f} MB + logits_per_text {B * K * N * 4 / 1e6:.0f} MB materialised)", flush=True)


if __name__ == "__main__":
    main()
