#!/usr/bin/env python3
"""Train-step timing of other CLIP geometries on the HIP path (BASELINE config 5's ViT-L/14 at 336 px, ViT-B/16), in
bf16 or with the fp8 operand path (--fp8 / --fp8=3).  Not the headline metric: bench.py stays on ViT-B/32.

    python tools/bench_arch.py vit_l14_336 [batch] [--fp8[=bits]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_event_amd import synthetic as S, distributed as D
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.model import build_model
from clip_event_amd.optim import FusedAdam

ARCH = {   # default batch, nominal fwd+bwd FLOP per pair
    "vit_b16": (128, 3 * 2 * (17.58e9 + 2.9798e9)),
    "vit_l14_336": (32, 1185.7e9),
}
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
fp8 = 0
for a in sys.argv[1:]:
    if a.startswith("--fp8"):            # --fp8 (forward GEMMs) / --fp8=3 (forward + input-gradient GEMMs)
        fp8 = int(a.split("=")[1]) if "=" in a else 1
name = argv[0] if argv else "vit_l14_336"
B, flop_pair = ARCH[name]
if len(argv) > 1:
    B = int(argv[1])
dev = torch.device("cuda", 0)
t0 = time.time()
model = S.synthetic_model(name, seed=0).to(dev)
model.fp8 = fp8
print(f"{name}: built in {time.time()-t0:.1f}s, {sum(p.numel() for p in model.parameters())/1e6:.0f} M parameters, "
      f"{model.visual.patch_num ** 2 + 1} image tokens", flush=True)
crit = CriterionContrastive("ce")
opt = FusedAdam(model, lr=1e-6)
img = S.synthetic_images(B, model.visual.input_resolution, seed=999).to(dev)
txt = S.synthetic_tokens(B, 77, 49408, seed=999).to(dev)
yi, yt, ip = D.global_labels(B, 1, 0, True, device=dev, rank_=0)
for _ in range(2):
    ld = train_step(model, crit, opt, img, txt, yi, yt, ip)
torch.cuda.synchronize()
N = 5
t0 = time.perf_counter()
for _ in range(N):
    ld = train_step(model, crit, opt, img, txt, yi, yt, ip)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
loss = float(sum(v.detach() for v in ld.values()))
print(f"{name} B={B} fp8={fp8}: {dt*1e3:.1f} ms/step = {B/dt:,.0f} pairs/s, loss {loss:.4f}, "
      f"{B/dt*flop_pair/2.5e15*100:.1f} % of bf16 peak on nominal FLOPs")

# per-class breakdown (single stream so that the HIP-event intervals are not stretched by overlap)
import ctypes
from clip_event_amd._lib import lib
cl = lib()
cl.ce_profile_class_name.restype = ctypes.c_char_p
model.tower_streams = False
for _ in range(2):
    train_step(model, crit, opt, img, txt, yi, yt, ip)
torch.cuda.synchronize()
cl.ce_profile_enable(1)
train_step(model, crit, opt, img, txt, yi, yt, ip)
torch.cuda.synchronize()
NC = int(cl.ce_profile_num_classes())
buf = (ctypes.c_double * (NC * 4))()
cl.ce_profile_collect(buf, NC)
cl.ce_profile_enable(0)
rows = []
for c in range(NC):
    cnt, ms, fl, by = buf[c * 4:(c + 1) * 4]
    if cnt > 0:
        rows.append((ms, cl.ce_profile_class_name(c).decode(), cnt, fl / (ms * 1e-3) / 1e12 if ms else 0, by / (ms * 1e-3) / 1e9 if ms else 0))
for ms, n, cnt, tf, gb in sorted(rows, reverse=True):
    print(f"  {n:28s} {cnt:5.0f} launches {ms:8.3f} ms  {tf:7.1f} TF/s {gb:8.1f} GB/s")
