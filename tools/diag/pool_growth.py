#!/usr/bin/env python3
"""Does the merged config-4 step return its tower workspaces to the pool?  Allocated device memory and pool contents per step."""
import sys, time, torch
sys.path.insert(0, ".")
from clip_event_amd import synthetic as S, distributed as D
from clip_event_amd.engine import train_step
from clip_event_amd.losses import CriterionAlignment, CriterionContrastive
from clip_event_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
B, K = 64, 5
m = S.synthetic_model("vit_b32", seed=0).to(dev)
m.set_hyps(True, True, False)
opt = FusedAdam(m, lr=1e-6, max_norm=1.0)
img = S.synthetic_images(B, 224, seed=1).to(dev)
txt = S.synthetic_tokens(B * K, 77, 49408, seed=2).to(dev)
yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=dev, rank_=0)
obj, obj_num, ent, ent_num = S.synthetic_entities(B, 224, 77, 49408, seed=7)
boxes = S.synthetic_bboxes(B, seed=8)
desc = [t.to(dev) for t in S.synthetic_role_texts(boxes, 77, 49408, seed=9)]
lab = [t.to(dev) for t in S.synthetic_role_texts(boxes, 77, 49408, seed=10)]
kw = dict(criterion_ot=CriterionAlignment(), object_vec=obj.to(dev), entitytxt_vec=ent.to(dev), object_num=obj_num.to(dev),
          entitytxt_num=ent_num.to(dev), train_arg="desc", bboxs=boxes, bbox_desc_vec=desc, bbox_label_vec=lab)
crit = CriterionContrastive("ce")
for it in range(14):
    t0 = time.perf_counter()
    train_step(m, crit, opt, img, txt.clone(), yi, yt, ip, **kw)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    pool = {k[0]: (len(v), k[1] >> 20) for k, v in m._pool.free.items()}
    print(f"step {it}: host {host * 1e3:6.1f} ms, allocated {torch.cuda.memory_allocated() >> 20} MiB, reserved {torch.cuda.memory_reserved() >> 20} MiB, pool {pool}")
