import sys, time, contextlib
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import clip_oracle as O
from clip_event_amd import synthetic as S
from tests.hostile import hostile_state, hostile_tokens
torch.set_num_threads(8)
def rel(a,b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
def cos(a,b):
    a,b=a.double().flatten(),b.double().flatten(); return float(a@b/(a.norm()*b.norm()+1e-30))
def run(cfg, B, seed, tag):
    sd = hostile_state(O.init_params(cfg, seed), cfg, seed=seed)
    img = S.synthetic_images(B, cfg.image_resolution, seed=31)
    txt = hostile_tokens(B, cfg.context_length, cfg.vocab_size, seed=32)
    y = torch.arange(B)
    t0=time.time()
    ld32, g32, (li32, lt32) = O.loss_and_grads(sd, cfg, img, txt, y, y, y, True)
    print(tag, "fp32 done", time.time()-t0, {k: float(v) for k,v in ld32.items()}, "logits range", float(li32.abs().max()))
    # residual stream magnitude
    for s16 in (False, True):
        with (O.stream_f16() if s16 else contextlib.nullcontext()):
            ld, g, (li, lt) = O.loss_and_grads(sd, cfg, img, txt, y, y, y, True, bf16=True)
            fi = O.encode_image(sd, cfg, img, bf16=True); ft = O.encode_text(sd, cfg, txt, bf16=True)
        fi32 = O.encode_image(sd, cfg, img); ft32 = O.encode_text(sd, cfg, txt)
        rels=[]; worst=(1,None)
        for k in g32:
            if g32[k] is None or float(g32[k].norm())==0: continue
            c=cos(g[k],g32[k]); rels.append(rel(g[k],g32[k]))
            if c<worst[0]: worst=(c,k)
        tn = lambda G: float(sum((v.double()**2).sum() for v in G.values() if v is not None)**0.5)
        print(tag, "stream16" if s16 else "stream32", "feat rel img %.3e txt %.3e"%(rel(fi,fi32),rel(ft,ft32)),
              "logit max|d| %.3f"%float((li-li32).abs().max()), "loss d %.4f %.4f"%(abs(float(ld['loss_i']-ld32['loss_i'])),abs(float(ld['loss_t']-ld32['loss_t']))),
              "worst cos %.5f %s"%worst, "median rel %.4f max %.4f"%(np.median(rels),max(rels)), "norm ratio %.4f"%(tn(g)/tn(g32)))
cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2)
run(cfg, 6, 11, "tiny")
if len(sys.argv)>1:
    run(O.VIT_B32, 8, 0, "vitb32")
