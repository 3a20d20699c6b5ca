"""Data-parallel pieces of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference only wraps the model in DistributedDataParallel (train.py:222-225) and computes
InfoNCE over each rank's local batch (engine.py:48-53); its gradient-preserving
``gather_tensors`` helper (utils.py:192-206) has no caller.  BASELINE.json's north star asks for
the *global*-batch loss, so this module provides

* ``gather_features`` / ``gather_feature_pair``: all-gather in rank order with a reduce-scatter(sum) backward, which makes
  the W-rank gradient equal to the single-process gradient on the concatenated batch once the
  parameter gradients are averaged;
* ``global_labels``: the label/index layout of dataset_voa.py:615-663 offset by ``rank*B``;
* ``GradSync``: mean all-reduce of the model's flat gradient buffer in a few pieces per tower, each
  launched as soon as the backward has enqueued the blocks it covers, so it overlaps the rest;
* ``reduce_dict``: utils.py:136-160 (logging only).

Everything here is device-agnostic (the CPU tests run it on gloo with world_size 2).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


# Rehearsal switch: with one rank, still issue every collective of the data-parallel path (they are identities at
# world size 1).  Lets the exact multi-GPU code path -- RCCL calls from the towers' side streams and from autograd's
# thread, gradient pieces, pending handles -- run on a one-GPU box (bench.py with CE_FORCE_COLLECTIVES=1).
FORCE_COLLECTIVES = os.environ.get("CE_FORCE_COLLECTIVES", "0") == "1"


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def active() -> bool:
    """True when the step must go through the collectives: more than one rank, or the rehearsal switch."""
    return is_dist() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


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
    if not active():
        return x
    return _AllGatherFn.apply(x)


def gather_feature_pair(fi: torch.Tensor, ft: torch.Tensor):
    """Both feature matrices of a step in ONE all-gather (and one reduce-scatter in the backward): the exchange is
    latency-bound (a few hundred KB per rank), so two collectives cost twice one.  ``fi`` [B, E], ``ft`` [B*K, E] ->
    ``(fi_all [W*B, E], ft_all [W*B*K, E])`` in rank order."""
    if not active():
        return fi, ft
    W = world_size()
    B, n = fi.shape[0], ft.shape[0]
    both = gather_features(torch.cat([fi, ft], dim=0)).view(W, B + n, fi.shape[1])
    return both[:, :B].reshape(W * B, -1), both[:, B:].reshape(W * n, -1)


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
    """Mean all-reduce of the flat gradient buffer, bucketed per tower and, inside a tower, per group of
    residual blocks.

    ``model.grad_sync`` is called by each tower's backward: after every layer range (``upto_layer``: the blocks
    down to that one are final; their gradients are a contiguous prefix of the tower's range, model._prepare) and
    once more when the whole tower, embeddings included, has been enqueued.  Each piece's all-reduce is issued
    asynchronously (RCCL runs it on its own stream, ordered after the launches enqueued so far) and overlaps the
    rest of the backward.  ``finish()`` reduces what is left (logit_scale) and waits for everything before the
    optimiser runs.  xGMI is point-to-point, so a ring all-reduce of the 600 MB buffer costs milliseconds: only
    the last piece (lowest blocks + input embeddings of the tower that finishes last) stays exposed."""

    def __init__(self, model, pieces_per_tower: Optional[int] = None):
        self.model = model
        if pieces_per_tower is None:
            pieces_per_tower = int(os.environ.get("CE_GRAD_PIECES", "3"))
        self.pieces = max(1, int(pieces_per_tower))
        self.pending = []
        self.done = set()
        self.progress = {}
        model.grad_sync = self            # callable: (model, tower, upto_layer=None); also queried for layer_cuts

    def layer_cuts(self, tower: str, layers: int):
        """Blocks at which a tower's backward pauses to hand over gradients: ``pieces`` roughly equal groups."""
        if not active() or self.pieces < 2 or tower in self.done:
            return []
        cuts = sorted({(layers * k) // self.pieces for k in range(1, self.pieces)}, reverse=True)
        return [c for c in cuts if 0 < c < layers]

    def _reduce_range(self, a: int, b: int, async_op: bool):
        if b <= a:
            return
        buf = self.model._flat_grad[a:b]
        W = world_size()
        if dist.get_backend() == "gloo":
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            buf.div_(W)
        else:
            h = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=async_op)
            if async_op:
                self.pending.append(h)

    def _reduce(self, name: str, async_op: bool):
        a, b = self.model._ranges[name]
        self._reduce_range(a, b, async_op)

    def _on_tower(self, model, name: str, upto_layer: Optional[int] = None):
        if not active() or name in self.done:
            return
        # a tower that runs several passes per step (sim_entity, region branch) is reduced again at finish()
        a, b = model._ranges[name]
        start = self.progress.get(name, a)
        end = b if upto_layer is None else model._layer_end[name][upto_layer]
        self._reduce_range(start, end, async_op=True)
        self.progress[name] = end
        if upto_layer is None:
            self.done.add(name)

    __call__ = _on_tower

    def finish(self, passes_per_tower: int = 1):
        if not active():
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
                    a, b = self.model._ranges[name]
                    self._reduce_range(self.progress.get(name, a), b, async_op=False)
        self._reduce("head", async_op=False)
        for h in self.pending:
            h.wait()
        self.pending = []
        self.done = set()
        self.progress = {}
