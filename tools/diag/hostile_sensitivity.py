import sys, time, contextlib
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import clip_oracle as O
from clip_event_amd import synthetic as S
from tests.hostile import hostile_state, hostile_tokens
torch.set_num_threads(8)
def rel(a,b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
cfg=O.VIT_B32; B=8
sd = hostile_state(O.init_params(cfg, 0), cfg, seed=0)
txt = hostile_tokens(B, cfg.context_length, cfg.vocab_size, seed=32)
y = torch.arange(B)
for iseed in (31, 41, 51):
    img = S.synthetic_images(B, cfg.image_resolution, seed=iseed)
    _, g32, _ = O.loss_and_grads(sd, cfg, img, txt, y, y, y, True)
    for s16 in (False, True):
        with (O.stream_f16() if s16 else contextlib.nullcontext()):
            _, g, _ = O.loss_and_grads(sd, cfg, img, txt, y, y, y, True, bf16=True)
        errs = sorted(((rel(g[k], g32[k]), k) for k in g32 if g32[k] is not None and float(g32[k].norm())>0), reverse=True)
        print(iseed, "s16" if s16 else "s32", [(round(e,3), k.replace('transformer.resblocks.','b').replace('visual.','v.')) for e,k in errs[:5]], flush=True)
# structure of the fragile gradient: share of the outlier channels in the ln gain gradient
k='visual.transformer.resblocks.11.ln_1.weight'
g=g32[k]; top=g.abs().topk(5)
print(k, "norm", float(g.norm()), "top5 entries", top.values.tolist(), top.indices.tolist())
