#!/usr/bin/env python3
"""Where do the small torch copy / fill / elementwise launches of a training step come from?  torch.profiler (CPU side,
with stacks) over two steps of the bench's default workload; aten ops grouped by the innermost clip_event_amd frame."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

from clip_event_amd import distributed as D
from clip_event_amd import synthetic as S
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionContrastive
from clip_event_amd.optim import FusedAdam


def main():
    dev = torch.device("cuda", 0)
    B = 256
    model = S.synthetic_model("vit_b32", seed=0).to(dev)
    crit = CriterionContrastive("ce")
    opt = FusedAdam(model, lr=1e-6, weight_decay=0.0, max_norm=1.0)
    img = S.synthetic_images(B, 224, seed=999).to(dev)
    txt = S.synthetic_tokens(B, 77, 49408, seed=999).to(dev)
    yi, yt, ip = D.global_labels(B, 1, 0, True, device=dev, rank_=0)

    def step():
        return train_step(model, crit, opt, img, txt.clone(), yi, yt, ip)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.name in ("aten::empty", "aten::empty_strided", "aten::view", "aten::as_strided",
                                                            "aten::reshape", "aten::slice", "aten::select", "aten::detach",
                                                            "aten::alias", "aten::t", "aten::transpose", "aten::expand",
                                                            "aten::unsqueeze", "aten::squeeze", "aten::_unsafe_view",
                                                            "aten::empty_like", "aten::resize_", "aten::set_", "aten::lift_fresh",
                                                            "aten::is_pinned", "aten::item", "aten::_local_scalar_dense"):
            continue
        frame = next((f for f in (ev.stack or []) if "clip_event_amd" in f or "find_copies" in f), "?")
        agg[(ev.name, frame.strip()[-100:])] += 1
    for (name, frame), n in agg.most_common(45):
        print(f"{n / 2:6.1f}/step  {name:22s} {frame}")


if __name__ == "__main__":
    main()
