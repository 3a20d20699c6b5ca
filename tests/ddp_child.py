#!/usr/bin/env python3
"""One rank of the W>1 parity check of the HIP model (started by tests/test_ddp_gpu.py; gloo backend, every rank
on cuda:0).  Each rank runs ``engine.train_step`` on its shard through the real ``GradSync`` (pieces, pass
counting, finish); rank 0 then recomputes the step in a single process on the CONCATENATED batch inside
``distributed.local_only()`` (the reference never gathers, engine.py:48-53 / utils.py:192-206, so
the concatenated-batch run of the same path is the oracle) and compares the rank-mean losses and every parameter
gradient.

Per-sample losses that are SUMS over the local batch (loss_ot model_clip.py:707, the region losses :456-488) are
divided by W by the gradient mean, exactly as under the reference's DDP: the single-process reference weights
them by 1/W.

    CASE=k5 RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python tests/ddp_child.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


class _GradOnly:
    """Optimizer stand-in: the step under test ends with the averaged gradients."""

    def __init__(self, model):
        self.model = model

    def zero_grad(self):
        self.model.zero_grad()

    def step(self):
        pass


def wrapper_case(rank, W, dev, sd, cfg, B, img_all, txt_all, crit):
    """The reference's loop verbatim (engine.py:48-53, :87-90) on ``distributed.DistributedDataParallel(model)``:
    forward through the wrapper, LOCAL-batch criterion, ``backward()`` -- the gradients must be the rank mean when it
    returns (no explicit finish), i.e. equal the single-process gradient of the mean of the per-shard losses."""
    from clip_event_amd import distributed as D
    from clip_event_amd.model import build_model
    m = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
    model = D.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    y = torch.arange(B, device=dev)
    img, txt = img_all[rank * B:(rank + 1) * B].to(dev), txt_all[rank * B:(rank + 1) * B].to(dev)
    for _ in range(2):
        m.zero_grad()
        li, lt = model(img, txt)
        ld = crit(li, lt, y, y, index_pos=y, constrastive_overbatch=model.module.constrastive_overbatch)
        sum(ld.values()).backward()
    torch.cuda.synchronize()
    g = m._flat_grad.detach().clone()
    ok = True
    if rank == 0:
        with D.local_only():
            m1 = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
            m1.zero_grad()
            total = 0
            for r in range(W):
                li, lt = m1(img_all[r * B:(r + 1) * B].to(dev), txt_all[r * B:(r + 1) * B].to(dev))
                ld1 = crit(li, lt, y, y, index_pos=y)
                total = total + sum(ld1.values()) / W
            total.backward()
            torch.cuda.synchronize()
        rel = float((g - m1._flat_grad).norm() / m1._flat_grad.norm())
        print(f"[wrapper] gradient after backward() vs mean of per-shard gradients: rel-L2 {rel:.2e}", flush=True)
        ok = rel < 2e-3
        print(f"[wrapper] {'OK' if ok else 'FAILED'}", flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


def sharded_case(rank, W, dev, sd, B, shard, crit):
    """DESIGN 5 lever 2 on the HIP path: three `train_step`s with `GradSync(sharded=True)` (reduce-scattered pieces, clip + Adam on
    the own shards, all-gather of the masters) against the same steps with the all-reduce + replicated update.  Adam with
    eps = 1 so that the update is smooth in the gradient (the atomics' last-bit noise is otherwise amplified to 2 lr per
    element, tests/test_model_gpu.py::test_deferred_text_update_equals_the_one_launch_update)."""
    from clip_event_amd import distributed as D
    from clip_event_amd.engine import train_step
    from clip_event_amd.model import build_model
    from clip_event_amd.optim import FusedAdam
    args, kw = shard(rank * B, (rank + 1) * B, rank, W)
    out = {}
    for mode in (False, True):
        m = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
        m.set_hyps(True, False, False)
        sync = D.GradSync(m, sharded=mode)
        assert (sync.plan is not None) == mode
        opt = FusedAdam(m, lr=0.1, eps=1.0, max_norm=1.0)
        for it in range(3):
            train_step(m, crit, opt, *args, grad_sync=sync, **kw)
        torch.cuda.synchronize()
        if mode:
            try:
                opt.state_dict()
                raise AssertionError("state_dict() of sharded moments did not refuse")
            except RuntimeError as e:
                assert "consolidate" in str(e)
            D.consolidate(m, opt)
            opt.state_dict()
        with torch.no_grad():
            li, lt = m(args[0], args[1])
        out[mode] = (m._flat.detach().clone(), opt.m.clone(), opt.v.clone(), float(opt.grad_norm()), li.float().clone())
    ok = True
    for i, what in enumerate(("masters", "exp_avg", "exp_avg_sq")):
        a, b = out[True][i], out[False][i]
        rel = float((a - b).norm() / b.norm())
        print(f"[sharded] rank {rank} {what}: sharded vs replicated rel-l2 {rel:.3e}", flush=True)
        ok &= rel < (1e-4 if i == 0 else 2e-2)
    print(f"[sharded] rank {rank} grad norm {out[True][3]:.6f} vs {out[False][3]:.6f}", flush=True)
    ok &= abs(out[True][3] - out[False][3]) <= 1e-2 * abs(out[False][3])      # (third step: bf16-level noise of two steps behind it)
    rel = float((out[True][4] - out[False][4]).norm() / out[False][4].norm())
    print(f"[sharded] rank {rank} logits after the steps: rel-l2 {rel:.3e}", flush=True)
    ok &= rel < 2e-2
    # every rank holds the same masters
    mx = out[True][0].clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    ok &= bool(torch.equal(mx, out[True][0]))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"[sharded] {'OK' if float(flag) == 1.0 else 'FAILED'}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


def main():
    rank, W = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    case = os.environ.get("CASE", "k1")
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from oracle import clip_oracle as O
    from clip_event_amd import synthetic as S, distributed as D
    from clip_event_amd.engine import train_step, contrastive_step_losses
    from clip_event_amd.losses import CriterionAlignment, CriterionContrastive
    from clip_event_amd.model import build_model

    cfg = O.ClipConfig(64, 64, 4, 128, 32, 20, 512, 128, 2, 3)
    B = 3
    K = 5 if case in ("k5", "all") else (2 if case == "region" else 1)
    align = case in ("align", "all")
    region = {"region": "desc_type_text", "all": "desc"}.get(case)
    sd = O.init_params(cfg, 11)
    N = W * B
    img_all = S.synthetic_images(N, cfg.image_resolution, seed=5)
    txt_all = S.synthetic_tokens(N * K, cfg.context_length, cfg.vocab_size, seed=6, min_len=2)
    obj, obj_num, ent, ent_num = S.synthetic_entities(N, cfg.image_resolution, cfg.context_length, cfg.vocab_size, seed=7,
                                                      max_objects=2, max_entities=3)
    boxes = S.synthetic_bboxes(N, seed=8, max_roles=3)
    desc = S.synthetic_role_texts(boxes, cfg.context_length, cfg.vocab_size, seed=9)
    lab = S.synthetic_role_texts(boxes, cfg.context_length, cfg.vocab_size, seed=10)
    crit, crit_ot = CriterionContrastive("ce"), CriterionAlignment()

    def shard(lo, hi, r, nranks):
        kw = {}
        if align:
            kw.update(criterion_ot=crit_ot, object_vec=obj[lo:hi].to(dev), entitytxt_vec=ent[lo:hi].to(dev),
                      object_num=obj_num[lo:hi].to(dev), entitytxt_num=ent_num[lo:hi].to(dev))
        if region:
            kw.update(train_arg=region, bboxs=boxes[lo:hi], bbox_desc_vec=desc[lo:hi], bbox_label_vec=lab[lo:hi])
        yi, yt, ip = D.global_labels(hi - lo, 1, K - 1, True, device=dev, rank_=r)
        return (img_all[lo:hi].to(dev), txt_all[lo * K:hi * K].to(dev), yi, yt, ip), kw

    if case == "wrapper":
        return wrapper_case(rank, W, dev, sd, cfg, B, img_all, txt_all, crit)
    if case == "sharded":
        return sharded_case(rank, W, dev, sd, B, shard, crit)

    # ---- the W-rank step: real GradSync, every collective of the path ----
    m = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
    m.set_hyps(True, align, False)
    sync = D.GradSync(m)
    args, kw = shard(rank * B, (rank + 1) * B, rank, W)
    for it in range(2):           # twice: the second step proves the per-step bookkeeping resets
        ld = train_step(m, crit, _GradOnly(m), *args, grad_sync=sync, **kw)
    torch.cuda.synchronize()
    assert not sync.pending and not sync.dirty and sync.expected == {"visual": 0, "text": 0}
    g = m._flat_grad.detach().clone()
    red = D.reduce_dict({k: v.detach() for k, v in ld.items()})
    # every rank must hold the same averaged gradient
    gmax = g.clone()
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
    assert torch.equal(gmax, g) or float((gmax - g).abs().max()) == 0.0, "ranks disagree on the averaged gradient"

    ok = True
    if rank == 0:
        with D.local_only():
            m1 = build_model({k: v.clone() for k, v in sd.items()}).to(dev)
            m1.set_hyps(True, align, False)
            (im, tx, yi, yt, ip), kw1 = shard(0, N, 0, 1)
            m1.zero_grad()
            ld1 = contrastive_step_losses(m1, crit, im, tx, yi, yt, ip, train_arg=kw1.get("train_arg"),
                                          bboxs=kw1.get("bboxs"), bbox_desc_vec=kw1.get("bbox_desc_vec"),
                                          bbox_label_vec=kw1.get("bbox_label_vec"))
            if align:
                fi, ft = m1.sim_entity(kw1["object_vec"], kw1["entitytxt_vec"])
                ld1.update(crit_ot(ft, fi, kw1["entitytxt_num"], kw1["object_num"]))
            total = sum(v if k in ("loss_i", "loss_t") else v / W for k, v in ld1.items())
            total.backward()
            torch.cuda.synchronize()
        for k in ld1:
            want = float(ld1[k]) if k in ("loss_i", "loss_t") else float(ld1[k]) / W
            got = float(red[k])
            print(f"[{case}] {k}: W-rank mean {got:.5f} vs single-process {want:.5f}", flush=True)
            if abs(got - want) > 3e-3 * max(1.0, abs(want)):
                ok = False
        worst_cos, worst_rel, worst_name = 1.0, 0.0, ""
        for n, p in m1.named_parameters():
            o = m1._offsets[n]
            a = g[o:o + p.numel()].double()
            b = m1._flat_grad[o:o + p.numel()].double()
            if float(b.norm()) == 0.0:
                if float(a.norm()) != 0.0:
                    ok = False
                    print(f"[{case}] {n}: reference gradient is zero, W-rank is not", flush=True)
                continue
            cos = float(a @ b / (a.norm() * b.norm()))
            rel = float((a - b).norm() / b.norm())       # scale errors: a piece averaged twice / never / before a write
            if rel > worst_rel:
                worst_cos, worst_rel, worst_name = cos, rel, n
        print(f"[{case}] worst gradient rel-L2 {worst_rel:.2e} (cosine {worst_cos:.6f}) at {worst_name}", flush=True)
        # bf16 operands: the shard and the concatenated batch tile differently (tolerance as the packed/dense test)
        if worst_rel > 2e-3:
            ok = False
        print(f"[{case}] {'OK' if ok else 'FAILED'}", flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
