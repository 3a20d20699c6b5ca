#!/usr/bin/env python3
"""Throughput of the on-device preprocessing (ce_preprocess) on a batch of decoded 640x480 uint8 images."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd.preprocess import preprocess

B = 256
imgs = [torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, device="cuda:0") for _ in range(B)]
for _ in range(3):
    preprocess(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    out = preprocess(imgs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"{B} images 640x480 -> [B,3,224,224] fp32: {dt*1e3:.2f} ms/batch = {B/dt:,.0f} images/s "
      f"({B*480*640*3/dt/1e9:.1f} GB/s of source bytes)")
