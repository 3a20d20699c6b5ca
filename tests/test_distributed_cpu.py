"""CPU, gloo, world_size 2: the data-parallel path (feature all-gather with reduce-scatter
backward, rank-offset labels, per-rank row blocks, gradient averaging) reproduces the
single-process loss and gradient on the concatenated batch.

The towers are HIP-only, so the features here are a differentiable stand-in (a linear map of
per-sample inputs) and the logits/criterion math is the CPU oracle's; what is under test is
the product's distributed module."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, W, port, B, K, out, paired):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    torch.set_num_threads(1)
    from oracle import clip_oracle as O
    from clip_event_amd import distributed as D
    g = torch.Generator().manual_seed(7)
    E = 16
    Wi = torch.randn(8, E, generator=g, requires_grad=True)      # shared "tower" parameters
    Wt = torch.randn(8, E, generator=g, requires_grad=True)
    ls = torch.tensor(2.0, requires_grad=True)
    xi_all = torch.randn(W * B, 8, generator=g)
    xt_all = torch.randn(W * B * K, 8, generator=g)
    xi, xt = xi_all[rank * B:(rank + 1) * B], xt_all[rank * B * K:(rank + 1) * B * K]
    fi, ft = xi @ Wi, xt @ Wt
    fi_all, ft_all = D.gather_feature_pair(fi, ft) if paired else (D.gather_features(fi), D.gather_features(ft))
    lpi, _ = O.logits_from_features(fi, ft_all, ls, True)          # local image rows x all texts
    _, lpt = O.logits_from_features(fi_all, ft, ls, True)          # local text rows x all images
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, rank_=rank)
    ld = O.criterion_contrastive(lpi, lpt, yi, yt, ip, "ce")
    (ld["loss_i"] + ld["loss_t"]).backward()
    grads = [Wi.grad, Wt.grad, ls.grad.reshape(1)]
    for gr in grads:                                               # DDP-style mean of parameter gradients
        dist.all_reduce(gr)
        gr /= W
    red = D.reduce_dict({k: v.detach() for k, v in ld.items()})
    if rank == 0:
        torch.save({"Wi": Wi.grad, "Wt": Wt.grad, "ls": ls.grad, "loss_i": red["loss_i"], "loss_t": red["loss_t"]}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("paired", [True, False])
def test_global_batch_gradient_equals_single_process(tmp_path, paired):
    from oracle import clip_oracle as O
    W, B, K = 2, 3, 2
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(W, _free_port(), B, K, out, paired), nprocs=W, join=True)
    got = torch.load(out, weights_only=True)
    g = torch.Generator().manual_seed(7)
    E = 16
    Wi = torch.randn(8, E, generator=g, requires_grad=True)
    Wt = torch.randn(8, E, generator=g, requires_grad=True)
    ls = torch.tensor(2.0, requires_grad=True)
    xi_all = torch.randn(W * B, 8, generator=g)
    xt_all = torch.randn(W * B * K, 8, generator=g)
    lpi, lpt = O.logits_from_features(xi_all @ Wi, xt_all @ Wt, ls, True)
    yi, yt, ip = O.build_labels(W * B, 1, K - 1, True)
    ld = O.criterion_contrastive(lpi, lpt, yi, yt, ip, "ce")
    (ld["loss_i"] + ld["loss_t"]).backward()
    assert torch.allclose(got["loss_i"], ld["loss_i"].detach(), atol=1e-6)
    assert torch.allclose(got["loss_t"], ld["loss_t"].detach(), atol=1e-6)
    assert torch.allclose(got["Wi"], Wi.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(got["Wt"], Wt.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(got["ls"], ls.grad, atol=1e-5, rtol=1e-4)


# ---------------------------------------------------------------------------------------------------------------
# GradSync itself (layer_cuts, pass counting, pieces, finish) on gloo with a stand-in for the flat gradient buffer

class _FlatStandIn:
    """What GradSync touches of the model: the flat gradient buffer, the per-group ranges and the per-block
    prefix ends (model.CLIP._prepare)."""
    LAYERS = 4

    def __init__(self, tail=3):
        blk, headn = 6, 2
        self._ranges, self._layer_end = {}, {}
        off = 0
        self._ranges["head"] = (off, off + 1)
        off += 1
        for tower in ("visual", "text"):
            start = off
            off += headn
            self._layer_end[tower] = {}
            for b in reversed(range(self.LAYERS)):
                off += blk
                self._layer_end[tower][b] = off
            off += tail
            self._ranges[tower] = (start, off)
        self._flat_grad = torch.zeros(off)
        self.grad_sync = None

    def backward_pass(self, tower, contribution):
        """Mirror of functional._tower_backward + the tower node's tail: accumulate ``contribution`` range by
        range, calling the hooks exactly where the HIP backward does."""
        gs = self.grad_sync
        a, b = self._ranges[tower]
        cuts = gs.layer_cuts(tower, self.LAYERS)
        pos = a
        for lo in list(cuts) + [0]:
            end = self._layer_end[tower][lo]
            self._flat_grad[pos:end] += contribution[pos - a:end - a]
            pos = end
            if lo > 0:
                gs(self, tower, upto_layer=lo)
        self._flat_grad[pos:b] += contribution[pos - a:b - a]
        gs(self, tower)


def _gradsync_worker(rank, W, port, passes, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from clip_event_amd import distributed as D
    m = _FlatStandIn()
    sync = D.GradSync(m, pieces_per_tower=3)
    results = []
    for step in range(2):                     # two steps: the bookkeeping must reset
        m._flat_grad.zero_()
        g = torch.Generator().manual_seed(100 * step + rank)
        contrib = {t: [torch.randn(m._ranges[t][1] - m._ranges[t][0], generator=g) for _ in range(passes[t])]
                   for t in ("visual", "text")}
        for t in ("visual", "text"):
            for _ in range(passes[t]):
                sync.note_forward(t)
        if passes.get("unnoted"):              # a backward whose forward was never reported: nothing eager, finish() reduces
            sync.expected["text"] = 0
        cuts_seen = {}
        # autograd order: the passes created last run first; the towers interleave
        for k in reversed(range(max(passes["visual"], passes["text"]))):
            for t in ("text", "visual"):
                if k < passes[t]:
                    cuts_seen.setdefault(t, []).append(len(sync.layer_cuts(t, m.LAYERS)))
                    m.backward_pass(t, contrib[t][k])
        m._flat_grad[0] += float(rank + 1)     # logit_scale
        sync.finish()
        sync.finish()                           # idempotent
        assert not sync.dirty and not sync.pending
        for t in ("visual", "text"):
            n = len(cuts_seen[t])
            # only the last pass through a tower may be cut into eagerly reduced pieces
            assert all(c == 0 for c in cuts_seen[t][:-1]), cuts_seen
            if not (passes.get("unnoted") and t == "text"):
                assert cuts_seen[t][-1] == 2, cuts_seen
        results.append((m._flat_grad.clone(), {t: torch.stack(contrib[t]).sum(0) for t in contrib}))
    gathered = [None] * W
    dist.all_gather_object(gathered, [(r[0], r[1]) for r in results])
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("passes", [{"visual": 1, "text": 1}, {"visual": 2, "text": 3},
                                    {"visual": 2, "text": 2, "unnoted": True}])
def test_gradsync_pieces_and_multi_pass_average(tmp_path, passes):
    """Every element of the flat gradient buffer ends as the rank mean of the summed per-pass contributions,
    whatever the number of passes per tower; a range is never reduced before its last write (a piece averaged
    early and written again would come out as (mean + later)/W, which this catches)."""
    W = 2
    out = str(tmp_path / "g.pt")
    mp.spawn(_gradsync_worker, args=(W, _free_port(), passes, out), nprocs=W, join=True)
    gathered = torch.load(out, weights_only=False)          # written by this test
    m = _FlatStandIn()
    for step in range(2):
        want = torch.zeros_like(m._flat_grad)
        for r in range(W):
            tot = gathered[r][step][1]
            for t in ("visual", "text"):
                a, b = m._ranges[t]
                want[a:b] += tot[t] / W
            want[0] += (r + 1) / W
        for r in range(W):
            assert torch.allclose(gathered[r][step][0], want, atol=1e-6), (step, r)


def test_local_only_switch():
    from clip_event_amd import distributed as D
    assert not D.active() and D.world_size() == 1 and D.rank() == 0
    with D.local_only():
        assert not D.active()
    assert D._LOCAL_ONLY == 0


def test_ddp_wrapper_surface_and_torch_ddp_is_refused():
    """train.py:222-225's call site: our wrapper exposes ``.module`` and forwards calls; the model refuses to run
    inside torch's DistributedDataParallel with a message that names the replacement."""
    import torch.nn.parallel as P
    from clip_event_amd import distributed as D
    from clip_event_amd.model import CLIP
    m = CLIP(32, 32, 1, 64, 16, 8, 64, 64, 1, 1)
    w = D.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    assert w.module is m and isinstance(m.grad_sync, D.GradSync)
    assert all(k.startswith("module.") for k in w.state_dict())
    P.DistributedDataParallel._active_ddp_module = object()
    try:
        with pytest.raises(RuntimeError, match="clip_event_amd.distributed.DistributedDataParallel"):
            m._note_pass("visual")
    finally:
        P.DistributedDataParallel._active_ddp_module = None


# ---- sharded optimiser step (DESIGN 5 lever 2): reduce-scatter + per-shard clip / Adam + all-gather of the masters ----------

def _adam_reference(p, g, m, v, coef, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    g = g * coef
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_(lr / (1 - b1 ** step) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))


def _sharded_worker(rank, W, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from clip_event_amd import distributed as D
    m = _FlatStandIn(tail=4)                                   # every piece splits into two equal shards
    n = m._flat_grad.numel()
    m._flat = torch.randn(n, generator=torch.Generator().manual_seed(5))
    sync = D.GradSync(m, pieces_per_tower=3, sharded=True)
    plan = sync.plan
    assert plan is not None and len(plan.pieces) == 6 and plan.head == (0, 1)
    owned = plan.owned(rank)
    mine = torch.zeros(n, dtype=torch.bool)
    for lo, hi in owned:
        mine[lo:hi] = True
    mine[0] = True
    mom, var = torch.zeros(n), torch.zeros(n)
    sumsq = torch.zeros(1)
    lr, max_norm = 0.05, 1.0
    grads = []
    for step in range(1, 4):
        m._flat_grad.zero_()
        g = torch.Generator().manual_seed(100 * step + rank)
        contrib = {t: torch.randn(m._ranges[t][1] - m._ranges[t][0], generator=g) for t in ("visual", "text")}
        for t in ("visual", "text"):
            sync.note_forward(t)
        for t in ("text", "visual"):
            m.backward_pass(t, contrib[t])
        m._flat_grad[0] += float(rank + 1)
        sync.finish()
        grads.append(m._flat_grad.clone())
        m._flat_grad[~mine] = float("nan")                     # what a real reduce-scatter leaves outside the own shards: nothing usable
        touched = []

        def sumsq_fn(lo, hi):
            sumsq.add_(m._flat_grad[lo:hi].square().sum())

        def adam_fn(lo, hi):
            coef = min(1.0, max_norm / (float(sumsq.sqrt()) + 1e-6))
            _adam_reference(m._flat[lo:hi], m._flat_grad[lo:hi], mom[lo:hi], var[lo:hi], coef, lr, step)
            touched.append((lo, hi))

        D.sharded_update(plan, m._flat, sumsq, sumsq_fn, adam_fn)
        assert sorted(touched) == sorted(owned + [plan.head])
        assert bool(torch.isfinite(m._flat).all())

    class _Opt:
        pass
    opt = _Opt()
    opt.m, opt.v, opt._moments_stale = mom, var, True
    before = mom.clone()
    D.consolidate(m, opt)
    assert not opt._moments_stale and torch.equal(mom[mine], before[mine])
    gathered = [None] * W
    dist.all_gather_object(gathered, (m._flat, mom, var, grads))
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_optimizer_step_equals_replicated_step(tmp_path):
    """Reduce-scattered gradient pieces + clip / Adam on the own shard of every piece + all-gather of the masters leave every rank
    with the parameters a replicated step (mean gradients, one clip + Adam over everything) produces, and `consolidate` the
    moments.  Non-owned gradient shards are poisoned before the update: nothing may read them."""
    W = 2
    out = str(tmp_path / "s.pt")
    mp.spawn(_sharded_worker, args=(W, _free_port(), out), nprocs=W, join=True)
    gathered = torch.load(out, weights_only=False)          # written by this test
    m = _FlatStandIn(tail=4)
    n = m._flat_grad.numel()
    p = torch.randn(n, generator=torch.Generator().manual_seed(5))
    mom, var = torch.zeros(n), torch.zeros(n)
    for step in range(1, 4):
        g = gathered[0][3][step - 1]                           # rank means, identical on both ranks (gloo all-reduces whole pieces)
        assert torch.allclose(g, gathered[1][3][step - 1], atol=1e-7)
        coef = min(1.0, 1.0 / (float(g.square().sum().sqrt()) + 1e-6))
        _adam_reference(p, g, mom, var, coef, 0.05, step)
    for r in range(W):
        assert torch.allclose(gathered[r][0], p, atol=1e-6), r
        assert torch.allclose(gathered[r][1], mom, atol=1e-6) and torch.allclose(gathered[r][2], var, atol=1e-6), r
