#!/usr/bin/env python3
"""How far ahead of the GPU does the host run?  Times the enqueue of N training steps (no synchronisation inside)
against the wall time including the final synchronise, and the host time of the main pieces of one step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import synthetic as S, distributed as D
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.model import build_model
from clip_event_amd.optim import FusedAdam

dev = torch.device("cuda", 0)
B = 256
model = S.synthetic_model("vit_b32", seed=0).to(dev)
crit = CriterionContrastive("ce")
opt = FusedAdam(model, lr=1e-6)
img = S.synthetic_images(B, 224, seed=999).to(dev)
txt = S.synthetic_tokens(B, 77, 49408, seed=999).to(dev)
yi, yt, ip = D.global_labels(B, 1, 0, True, device=dev, rank_=0)
for _ in range(3):
    train_step(model, crit, opt, img, txt, yi, yt, ip)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    train_step(model, crit, opt, img, txt, yi, yt, ip)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/N:.2f} ms/step, wall {1e3*(t2-t0)/N:.2f} ms/step, drain after last enqueue {1e3*(t2-t1):.2f} ms")
# host cost with an idle GPU (sync before each step): pure CPU time to enqueue one step
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    train_step(model, crit, opt, img, txt, yi, yt, ip)
    ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print("host-only enqueue of one step (GPU idle at start): %.2f ms" % (1e3 * min(ts)))
