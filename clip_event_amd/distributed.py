"""Data-parallel pieces of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference only wraps the model in DistributedDataParallel (train.py:222-225) and computes
InfoNCE over each rank's local batch (engine.py:48-53); its gradient-preserving
``gather_tensors`` helper (utils.py:192-206) has no caller.  BASELINE.json's north star asks for
the *global*-batch loss, so this module provides

* ``gather_features``: all-gather in rank order with a reduce-scatter(sum) backward, which makes
  the W-rank gradient equal to the single-process gradient on the concatenated batch once the
  parameter gradients are averaged;
* ``global_labels``: the label/index layout of dataset_voa.py:615-663 offset by ``rank*B``;
* ``GradSync``: mean all-reduce of the model's flat gradient buffer, one bucket per tower,
  launched as soon as that tower's backward has been enqueued so it overlaps the other tower;
* ``reduce_dict``: utils.py:136-160 (logging only).

Everything here is device-agnostic (the CPU tests run it on gloo with world_size 2).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


class _AllGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        W = dist.get_world_size()
        x = x.contiguous()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if dist.get_backend() == "gloo":          # rehearsal backend: list form
            dist.all_gather(list(out.chunk(W, dim=0)), x)
        else:
            dist.all_gather_into_tensor(out, x)
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        if dist.get_backend() == "gloo":          # gloo has no reduce_scatter: all-reduce + slice
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            r = dist.get_rank()
            out.copy_(g[r * ctx.n:(r + 1) * ctx.n])
        else:
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        return out


def gather_features(x: torch.Tensor) -> torch.Tensor:
    """[n, E] on every rank -> [W*n, E] in rank order; gradient flows back to every owner."""
    if world_size() == 1:
        return x
    return _AllGatherFn.apply(x)


def global_labels(batch: int, num_pos: int = 1, num_neg: int = 0, overbatch: bool = True, device=None,
                  rank_: Optional[int] = None):
    """Targets for this rank's rows against the all-gathered columns: image i of rank r is global
    image r*B+i, its positive description sits at column (r*B+i)*K (dataset_voa.py:619), its text
    rows point at image r*B+i (dataset_voa.py:652-655); ``index_pos`` selects the local positive
    rows (dataset_voa.py:658-663)."""
    K = num_pos + num_neg
    if num_pos != 1:
        raise RuntimeError("Only description_num_pos == 1 is laid out for constrative_loss == 'ce'")
    r = rank() if rank_ is None else rank_
    base = r * batch
    ar = torch.arange(batch, device=device)
    labels_per_image = (ar + base) * K if overbatch else torch.zeros(batch, dtype=torch.long, device=device)
    labels_per_text = (ar + base).unsqueeze(1).expand(batch, K).flatten()
    mask = torch.tensor([1] * num_pos + [0] * num_neg, device=device).repeat(batch)
    index_pos = torch.nonzero(mask).flatten()
    return labels_per_image, labels_per_text, index_pos


def reduce_dict(input_dict: Dict[str, torch.Tensor], average: bool = True) -> Dict[str, torch.Tensor]:
    """utils.py:136-160: all-reduce a dict of scalars (sorted keys), mean over ranks."""
    W = world_size()
    if W < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float().reshape(()) for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= W
        return {k: v for k, v in zip(names, values)}


class GradSync:
    """Mean all-reduce of the flat gradient buffer, bucketed per tower.

    ``model.grad_sync`` is called by each tower's backward right after its last launch was
    enqueued; the bucket's all-reduce is issued asynchronously (RCCL runs it on its own stream)
    and overlaps the other tower's backward.  ``finish()`` reduces the small head bucket and
    waits for everything before the optimiser runs."""

    def __init__(self, model):
        self.model = model
        self.pending = []
        self.done = set()
        model.grad_sync = self._on_tower

    def _reduce(self, name: str, async_op: bool):
        m = self.model
        a, b = m._ranges[name]
        if b <= a:
            return
        buf = m._flat_grad[a:b]
        W = world_size()
        if dist.get_backend() == "gloo":
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            buf.div_(W)
        else:
            h = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=async_op)
            if async_op:
                self.pending.append(h)

    def _on_tower(self, model, name: str):
        if world_size() < 2 or name in self.done:
            return
        # a tower that runs several passes per step (sim_entity, region branch) is reduced at finish()
        self.done.add(name)
        self._reduce(name, async_op=True)

    def finish(self, passes_per_tower: int = 1):
        if world_size() < 2:
            return
        if passes_per_tower > 1:
            for h in self.pending:
                h.wait()
            self.pending = []
            for name in ("visual", "text"):
                self._reduce(name, async_op=False)
        else:
            for name in ("visual", "text"):
                if name not in self.done:
                    self._reduce(name, async_op=False)
        self._reduce("head", async_op=False)
        for h in self.pending:
            h.wait()
        self.pending = []
        self.done = set()
