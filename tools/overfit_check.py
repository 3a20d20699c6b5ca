#!/usr/bin/env python3
"""End-to-end sanity of the training dynamics: ViT-B/32 on ONE fixed synthetic batch must drive the InfoNCE loss
towards zero (fused clip + Adam, warm-up cosine schedule, packed text tower, two-stream towers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import synthetic as S, distributed as D
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.model import build_model
from clip_event_amd.optim import FusedAdam, WarmupCosineLR

dev = torch.device("cuda", 0)
B, STEPS = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 150
model = S.synthetic_model("vit_b32", seed=0).to(dev)
crit = CriterionContrastive("ce")
opt = FusedAdam(model, lr=2e-5, weight_decay=0.0, max_norm=1.0)
sch = WarmupCosineLR(opt, max_iters=STEPS, warmup_epochs=10)
img = S.synthetic_images(B, 224, seed=1).to(dev)
txt = S.synthetic_tokens(B, 77, 49408, seed=2).to(dev)
yi, yt, ip = D.global_labels(B, 1, 0, True, device=dev, rank_=0)
first = last = None
for it in range(STEPS):
    ld = train_step(model, crit, opt, img, txt, yi, yt, ip)
    sch.step()
    if it % 10 == 0 or it == STEPS - 1:
        loss = float(sum(v.detach() for v in ld.values()))
        gn = float(opt.grad_norm())
        print(f"step {it:4d} loss {loss:8.4f} grad-norm {gn:8.3f} lr {opt.param_groups[0]['lr']:.2e}", flush=True)
        first = loss if first is None else first
        last = loss
        assert loss == loss, "NaN loss"
print(f"loss {first:.4f} -> {last:.4f}")
assert last < 0.25 * first, "the fixed batch was not fitted"
print("overfit check OK")
