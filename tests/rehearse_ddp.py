#!/usr/bin/env python3
"""Rehearsal of the W>1 path on ONE GPU (both ranks on cuda:0, gloo backend): the W-rank global-batch loss and
parameter gradients of the HIP path must equal the single-process run on the concatenated batch.

    CE_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29511 tests/rehearse_ddp.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    rank, W = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from oracle import clip_oracle as O
    from clip_event_amd import synthetic as S, distributed as D
    from clip_event_amd.engine import contrastive_step_losses
    from clip_event_amd.losses import CriterionContrastive
    from clip_event_amd.model import build_model
    cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2)
    B, K = 3, 2
    sd = O.init_params(cfg, 11)
    img_all = S.synthetic_images(W * B, 64, seed=5)
    txt_all = S.synthetic_tokens(W * B * K, 20, 512, seed=6, min_len=2)
    crit = CriterionContrastive("ce")

    def run(model, img, txt, yi, yt, ip):
        model.zero_grad(set_to_none=True)
        ld = contrastive_step_losses(model, crit, img.to(dev), txt.to(dev), yi.to(dev), yt.to(dev), ip.to(dev))
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        return ld, {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    m = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
    sync = D.GradSync(m)
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, rank_=rank)
    ld, g = run(m, img_all[rank * B:(rank + 1) * B], txt_all[rank * B * K:(rank + 1) * B * K], yi, yt, ip)
    sync.finish()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    red = D.reduce_dict({k: v.detach() for k, v in ld.items()})
    if rank == 0:
        # single-process reference on the concatenated batch (same HIP path, no process group involvement)
        m1 = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
        yi1, yt1, ip1 = D.global_labels(W * B, 1, K - 1, True, rank_=0)
        saved = D.world_size
        D.world_size = lambda: 1
        ld1, g1 = run(m1, img_all, txt_all, yi1, yt1, ip1)
        D.world_size = saved
        worst, worst_rel = 1.0, 0.0
        for n in g:
            a, b = g[n].double().flatten(), g1[n].double().flatten()
            if float(b.norm()) > 0:
                worst = min(worst, float(a @ b / (a.norm() * b.norm())))
                worst_rel = max(worst_rel, float((a - b).norm() / b.norm()))       # scale errors (a piece averaged twice / never)
        print(f"loss_i W-rank mean {float(red['loss_i']):.5f} vs single {float(ld1['loss_i']):.5f}; "
              f"loss_t {float(red['loss_t']):.5f} vs {float(ld1['loss_t']):.5f}; worst grad cosine {worst:.6f}, worst rel-l2 {worst_rel:.2e}", flush=True)
        assert abs(float(red["loss_i"]) - float(ld1["loss_i"])) < 2e-3 and abs(float(red["loss_t"]) - float(ld1["loss_t"])) < 2e-3
        assert worst > 0.999 and worst_rel < 3e-2
        print("rehearsal OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
