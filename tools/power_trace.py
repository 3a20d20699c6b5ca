#!/usr/bin/env python3
"""Sample rocm-smi (power, sclk, mclk, temperature) in a thread while the training step runs: is the step running into
the power limit?"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import synthetic as S, distributed as D
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.optim import FusedAdam

samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--csv"], capture_output=True,
                                 text=True, timeout=5).stdout
            samples.append((time.time(), out))
        except Exception as e:      # noqa: BLE001
            samples.append((time.time(), f"ERR {e}"))
        time.sleep(0.2)


dev = torch.device("cuda", 0)
model = S.synthetic_model("vit_b32").to(dev)
crit = CriterionContrastive("ce")
opt = FusedAdam(model, lr=1e-6)
B = 256
img = S.synthetic_images(B, 224, seed=999).to(dev)
txt = S.synthetic_tokens(B, 77, 49408, seed=999).to(dev)
yi, yt, ip = D.global_labels(B, 1, 0, True, device=dev, rank_=0)
for _ in range(3):
    train_step(model, crit, opt, img, txt, yi, yt, ip)
torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
t0 = time.time()
n = 0
while time.time() - t0 < 8.0:
    for _ in range(20):
        train_step(model, crit, opt, img, txt, yi, yt, ip)
    torch.cuda.synchronize()
    n += 20
t1 = time.time()
time.sleep(1.0)
stop = True
th.join()
print(f"{n} steps, {(t1-t0)/n*1e3:.2f} ms/step")
print("idle sample:\n", samples[0][1][:600])
mid = [s for s in samples if t0 + 1 < s[0] < t1 - 0.5]
print(f"{len(mid)} samples under load; first and last:")
for s in (mid[:1] + mid[-1:]):
    print(s[1][:600])
