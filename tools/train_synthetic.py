#!/usr/bin/env python3
"""The reference's training loop (train.py:101-250 / engine.py:24-102) on the drop-in API, end to end on synthetic
data: decoded uint8 images -> on-device preprocessing -> CLIP forward (hard-negative descriptions, per-batch labels as
dataset_voa.py builds them) -> CriterionContrastive -> fused clip + Adam -> warm-up cosine schedule -> checkpoint in
the reference's layout -> resume from it.  A smoke run of every host-side component together, not a benchmark."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from clip_event_amd import checkpoint, clip, distributed as D, synthetic as S
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.optim import build_lr_scheduler, build_optimizer
from clip_event_amd.preprocess import preprocess

CAPTIONS = list(S.ASCII_CAPTIONS)


def batch(rng, B, K, dev):
    sizes = [(int(rng.integers(240, 640)), int(rng.integers(240, 640))) for _ in range(B)]
    imgs = [torch.from_numpy(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).to(dev) for (w, h) in sizes]
    image = preprocess(imgs)                                                    # clip.py:62-69 on the GPU
    texts = [CAPTIONS[int(rng.integers(len(CAPTIONS)))] + f" number {int(rng.integers(1000))}" for _ in range(B * K)]
    text = clip.tokenize(texts)                                                 # clip.py:168-201: stays on the HOST, as the reference's loader
                                                                                # yields it -- train_step copies it and keeps the caption lengths
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=dev)                 # dataset_voa.py:605-664
    return image, text, yi, yt, ip


def main():
    dev = torch.device("cuda", 0)
    cfg = {"optimizer": "adam", "lr": 1e-5, "weight_decay": 0.0, "momentum": 0.9, "lr_scheduler": "warmup",
           "max_epoch": 40, "warmup_epoch": 4, "lr_steps": [], "lr_gamma": 0.1, "task": "clipevent"}
    B, K = 16, 3
    rng = np.random.default_rng(0)
    model = S.synthetic_model("vit_b32", seed=0).to(dev)
    model.set_hyps(constrastive_overbatch=True, alignment=False, multiattention=False)
    criterion = CriterionContrastive("ce")
    optimizer = build_optimizer(cfg, model)                                     # engine.py:129-151
    scheduler = build_lr_scheduler(cfg, optimizer, 0)                           # engine.py:154-176
    data = batch(rng, B, K, dev)                                                # one fixed batch: the loss must fall
    losses = []
    for it in range(20):
        ld = train_step(model, criterion, optimizer, *data, check_finite=True)
        scheduler.step()
        losses.append(float(sum(v.detach() for v in ld.values())))
    print("loss:", " ".join(f"{v:.3f}" for v in losses[::3]), "lr", optimizer.param_groups[0]["lr"])
    assert losses[-1] < 0.5 * losses[0]
    with tempfile.TemporaryDirectory() as d:
        path = checkpoint.save_model_on_master(model, d, cfg["task"], 20, 0.0, optimizer)      # engine.py:202-218
        model2, opt_state, begin_epoch, _ = checkpoint.load_checkpoint(path, device=dev)       # train.py:101-124
        optimizer2 = build_optimizer(cfg, model2)
        optimizer2.load_state_dict(opt_state)
        scheduler2 = build_lr_scheduler(cfg, optimizer2, begin_epoch)
    assert begin_epoch == 20 and abs(optimizer2.param_groups[0]["lr"] - optimizer.param_groups[0]["lr"]) < 1e-12
    a = train_step(model, criterion, optimizer, *data)
    b = train_step(model2, criterion, optimizer2, *data)
    la, lb = float(sum(v.detach() for v in a.values())), float(sum(v.detach() for v in b.values()))
    print(f"resumed run continues: loss {la:.5f} vs {lb:.5f}")
    assert abs(la - lb) < 1e-3 * max(1.0, abs(la))
    new = batch(rng, B, K, dev)                                                 # a fresh ragged batch goes through too
    ld = train_step(model2, criterion, optimizer2, *new)
    scheduler2.step()
    assert all(torch.isfinite(v) for v in ld.values())
    # a longer stretch of fresh batches with no host synchronisation in the loop: host-side caption lengths, the run-ahead
    # limit of engine.train_step, the asynchronous poll of the fp16-stream clamp counters (every 16 optimiser steps)
    import time
    batches = [batch(rng, 32, K, dev) for _ in range(8)]
    for it in range(4):                                                         # new batch size: workspaces are allocated here
        train_step(model2, criterion, optimizer2, *batches[it])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(96):
        ld = train_step(model2, criterion, optimizer2, *batches[it % 8])
        scheduler2.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 96
    assert all(torch.isfinite(v) for v in ld.values()) and model2.stream16_saturation() == (0, 0)
    print(f"96 steps at B = 32, K = {K}: {dt * 1e3:.2f} ms/step, clamp counters {model2.stream16_saturation()}")
    print("train_synthetic OK")


if __name__ == "__main__":
    main()
