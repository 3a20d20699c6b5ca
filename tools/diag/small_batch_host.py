"""Where does a small-batch step spend its time on the host?  B = 32, K = 3: host tokens / device tokens + lengths / device tokens."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clip_event_amd import synthetic as S, distributed as D
from clip_event_amd.engine import train_step
from clip_event_amd.functional import attach_lengths, host_lengths
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.optim import FusedAdam, WarmupCosineLR
dev = torch.device("cuda", 0)
B, K = 32, 3
m = S.synthetic_model("vit_b32", seed=0).to(dev)
crit = CriterionContrastive("ce")
opt = FusedAdam(m, lr=1e-6)
sched = WarmupCosineLR(opt, 1000, warmup_epochs=4)
img = S.synthetic_images(B, 224, seed=1).to(dev)
txt_h = S.synthetic_tokens(B * K, 77, 49408, seed=2)
txt_d = txt_h.to(dev)
lens = host_lengths(txt_h)
yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=dev)


def run(name, make, sched_step=False, n=60):
    for _ in range(5):
        train_step(m, crit, opt, img, make(), yi, yt, ip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        train_step(m, crit, opt, img, make(), yi, yt, ip)
        if sched_step:
            sched.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:44s} {(t2 - t0) / n * 1e3:7.2f} ms/step (host loop {(t1 - t0) / n * 1e3:.2f})")


run("device tokens, lengths attached", lambda: attach_lengths(txt_d.clone(), lens))
run("device tokens, no lengths (read-back)", lambda: txt_d.clone())
run("host tokens (pageable) -> train_step", lambda: txt_h.clone())
run("host tokens + scheduler.step()", lambda: txt_h.clone(), sched_step=True)
