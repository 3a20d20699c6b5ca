#!/usr/bin/env python3
"""Per-call time of FusedAdam.step() (sum of squares + update) on the ViT-B/32 parameter set: HIP events over 20 calls, and the
kernels' own times from the library's profile classes are not needed -- the two modes differ only in the update kernels."""
import os, sys, torch
sys.path.insert(0, ".")
from clip_event_amd import synthetic as S
from clip_event_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
m = S.synthetic_model("vit_b32", seed=0).to(dev)
opt = FusedAdam(m, lr=1e-6, weight_decay=0.0, max_norm=1.0)
opt.zero_grad()
m._flat_grad.normal_(0, 0.01)
for _ in range(3): opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): opt.step()
e1.record(); torch.cuda.synchronize()
print(f"CE_ADAM_TILES={os.environ.get('CE_ADAM_TILES', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per optimizer step (sumsq + update)")
