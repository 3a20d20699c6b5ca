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


_LOCAL_ONLY = 0          # depth of `local_only()` contexts in this process


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


class local_only:
    """Context manager: inside it this process computes as if no process group existed (``active()`` is False,
    ``world_size()`` 1, ``rank()`` 0, no collective is issued).  For single-process reference runs inside a
    multi-rank job -- e.g. rank 0 recomputing the concatenated batch in the W>1 parity tests -- and for strict
    reference-DDP equivalence of the loss (local InfoNCE)."""

    def __enter__(self):
        global _LOCAL_ONLY
        _LOCAL_ONLY += 1
        return self

    def __exit__(self, *exc):
        global _LOCAL_ONLY
        _LOCAL_ONLY -= 1
        return False


def active() -> bool:
    """True when the step must go through the collectives: more than one rank, or the rehearsal switch."""
    return _LOCAL_ONLY == 0 and is_dist() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def world_size() -> int:
    return dist.get_world_size() if (is_dist() and _LOCAL_ONLY == 0) else 1


def rank() -> int:
    return dist.get_rank() if (is_dist() and _LOCAL_ONLY == 0) else 0


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


def piece_cuts(layers: int, pieces: int):
    """Blocks at which a tower's gradient range is cut into ``pieces`` roughly equal groups of residual blocks (descending)."""
    cuts = sorted({(layers * k) // pieces for k in range(1, pieces)}, reverse=True)
    return [c for c in cuts if 0 < c < layers]


class ShardPlan:
    """Static partition of the flat parameter / gradient / moment buffers for the sharded optimiser step (lever 2):
    every tower's range is cut into the SAME pieces the gradient exchange uses (``piece_cuts``), every piece into ``W`` equal
    shards, and rank r owns shard r of every piece -- for good: its Adam moments never move.  A piece is the unit of the
    reduce-scatter (gradients, in place: the reduced shard lands where the rank's own part of the piece lies) and of the
    all-gather (updated fp32 masters, in place).  The layout puts every possible piece boundary on a multiple of 512
    elements (model._prepare), so W = 2, 4, 8 split a piece into whole 64-element groups."""

    def __init__(self, model, pieces: int, W: int):
        self.W = W
        self.pieces = []
        for tower in GradSync.TOWERS:
            a, b = model._ranges[tower]
            ends = model._layer_end[tower]
            bounds = [a] + [ends[c] for c in piece_cuts(len(ends), pieces)] + [b]
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                if hi > lo:
                    if (hi - lo) % W:
                        raise RuntimeError(f"a gradient piece of {hi - lo} elements does not split into {W} equal shards "
                                           "(the flat layout supports world sizes 2, 4 and 8 for the sharded optimiser)")
                    self.pieces.append((lo, hi))
        self.head = model._ranges["head"]

    def shard(self, piece, r: int):
        lo, hi = piece
        s = (hi - lo) // self.W
        return lo + r * s, lo + (r + 1) * s

    def owned(self, r: int):
        return [self.shard(p, r) for p in self.pieces]

    def split(self, a: int, b: int):
        """The pieces that make up [a, b) (whose ends are piece boundaries)."""
        out = [p for p in self.pieces if a <= p[0] and p[1] <= b]
        if sum(hi - lo for lo, hi in out) != b - a:
            raise RuntimeError(f"range [{a}, {b}) is not a union of gradient pieces")
        return out


def _reduce_scatter_mean(buf: torch.Tensor, W: int, r: int, async_op: bool):
    """Mean reduce-scatter of ``buf`` in place: rank r's shard of the result lands in ``buf[r s : (r + 1) s]``."""
    s = buf.numel() // W
    if dist.get_backend() == "gloo":              # rehearsal backend: no reduce-scatter -- all-reduce (a superset of the result)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.div_(W)
        return None
    return dist.reduce_scatter_tensor(buf[r * s:(r + 1) * s], buf, op=dist.ReduceOp.AVG, async_op=async_op)


def _all_gather_in_place(buf: torch.Tensor, W: int, r: int, async_op: bool = False):
    """``buf`` = W equal shards; every rank contributes its own (shard r) and receives the others in place."""
    s = buf.numel() // W
    if dist.get_backend() == "gloo":
        view = buf.view(torch.int16) if buf.dtype in (torch.bfloat16, torch.uint16) else buf      # (gloo moves bytes; it has no bf16 / u16)
        dist.all_gather(list(view.chunk(W)), view[r * s:(r + 1) * s].clone())
        return None
    return dist.all_gather_into_tensor(buf, buf[r * s:(r + 1) * s], async_op=async_op)


def sharded_update(plan: ShardPlan, params: torch.Tensor, sumsq: Optional[torch.Tensor], sumsq_fn, adam_fn):
    """The optimiser step over a reduce-scattered gradient buffer.  ``sumsq_fn(lo, hi)`` adds the sum of squares of the
    gradients ``[lo, hi)`` to ``sumsq`` (a device scalar, zeroed here; None: no clipping), ``adam_fn(lo, hi)`` applies
    clip + Adam to that range reading ``sumsq`` -- the caller's kernels (optim.FusedAdam: ce_sumsq / ce_adam_step).  Every
    rank updates its own shard of every piece (and the replicated head range, whose gradient was all-reduced), the global
    gradient norm costs one scalar all-reduce, and the flat fp32 parameter buffer ``params`` is completed by an all-gather
    in place, piece by piece.  (The masters, not the bf16 operand mirror, travel: the forward reads embeddings, LayerNorm
    parameters and biases in fp32, and a checkpoint can then be written from any rank; the wire carries what the all-reduce
    carried, each rank's Adam pass shrinks to 1 / W.  Only the Adam moments stay sharded: ``consolidate``.)"""
    W, r = world_size(), rank()
    owned = plan.owned(r)
    if sumsq is not None:
        sumsq.zero_()
        for lo, hi in owned:
            sumsq_fn(lo, hi)
        dist.all_reduce(sumsq, op=dist.ReduceOp.SUM)
        if plan.head[1] > plan.head[0]:
            sumsq_fn(*plan.head)
    for lo, hi in owned:
        adam_fn(lo, hi)
    if plan.head[1] > plan.head[0]:
        adam_fn(*plan.head)
    handles = [_all_gather_in_place(params[lo:hi], W, r, async_op=True) for lo, hi in plan.pieces]
    for h in handles:
        if h is not None:
            h.wait()


class GradSync:
    """Mean all-reduce of the flat gradient buffer, bucketed per tower and, inside a tower, per group of
    residual blocks -- the drop-in's counterpart of ``DistributedDataParallel``'s reducer (train.py:222-225).

    ``model.grad_sync`` is called by each tower's backward: after every layer range (``upto_layer``: the blocks
    down to that one are final; their gradients are a contiguous prefix of the tower's range, model._prepare) and
    once more when the whole tower, embeddings included, has been enqueued.  Each piece's all-reduce is issued
    asynchronously (RCCL runs it on its own stream, ordered after the launches enqueued so far on the issuing
    stream) and overlaps the rest of the backward.  xGMI is point-to-point, so a ring all-reduce of the 600 MB
    buffer costs milliseconds: only the last piece (lowest blocks + input embeddings of the tower that finishes
    last) stays exposed.

    Several passes through one tower in a step (``sim_entity`` with alignment, the role descriptions of the
    ``train_arg`` branch) all accumulate into the same gradient range.  The model reports every tower forward that
    will need a backward (``note_forward``); a range is handed to RCCL only during the LAST outstanding backward
    of its tower, and that backward's stream first waits on events recorded behind the gradient writes of the
    earlier passes -- so no reduction can start, on any stream, while a write to its range is still to come.
    Whatever was not reduced eagerly (a tower whose pass count is unknown, logit_scale) is reduced by
    ``finish()``, which runs as an autograd final callback at the end of ``backward()`` -- the gradients are
    averaged when ``backward()`` returns, as under DDP -- and may also be called explicitly (a second call is a
    no-op)."""

    TOWERS = ("visual", "text")
    plan = None          # ShardPlan of the sharded optimiser step, when switched on

    def __init__(self, model, pieces_per_tower: Optional[int] = None, auto_finish: bool = True, sharded: Optional[bool] = None):
        self.model = model
        if pieces_per_tower is None:
            pieces_per_tower = int(os.environ.get("CE_GRAD_PIECES", "3"))
        self.pieces = max(1, int(pieces_per_tower))
        self.auto_finish = auto_finish
        self.pending = []
        self._reset()
        # Sharded optimiser step (lever 2; CE_SHARDED_ADAM=1 or sharded=True): the pieces are reduce-SCATTERED, every
        # rank runs clip + Adam on its own shard of every piece (optim.FusedAdam.step sees ``self.plan``) and the fp32 masters
        # are all-gathered -- the all-reduce's bytes on the wire, 1 / W of the Adam pass per rank.  Only the Adam moments are
        # then current on their owner alone, until ``consolidate()`` gathers them (checkpoints).
        if sharded is None:
            sharded = os.environ.get("CE_SHARDED_ADAM", "0") == "1"
        self.plan = None
        if sharded and is_dist() and world_size() > 1:
            if getattr(model, "_ranges", None) is None and hasattr(model, "_ready"):
                model._ready()                   # the plan is cut from the flat layout
            self.plan = ShardPlan(model, self.pieces, world_size())
        model.grad_sync = self            # callable: (model, tower, upto_layer=None); also queried for layer_cuts
        # With a process group live, RCCL's channel kernels will sit on some CUs while the backward runs: hand the persistent
        # GEMMs' tiles out dynamically, so that a workgroup the dispatcher could not place does not hold a launch up for a whole
        # static tile list (15.3 -> 14.9 ms beside an 8-CU "hog", no cost without one).  CE_NT_DYNAMIC overrides.
        flat = getattr(model, "_flat_grad", None)
        on_gpu = flat.is_cuda if flat is not None else any(p.is_cuda for p in getattr(model, "parameters", lambda: [])())
        if active() and "CE_NT_DYNAMIC" not in os.environ and on_gpu:
            from ._lib import lib
            lib().ce_gemm_set_dynamic_tiles(1)

    def _reset(self):
        self.done = set()
        self.progress = {}
        self.expected = {t: 0 for t in self.TOWERS}      # tower forwards of this step that recorded a graph
        self.seen = {t: 0 for t in self.TOWERS}          # tower backwards completed this step
        self.fences = {t: [] for t in self.TOWERS}       # events behind the gradient writes of non-final passes
        self.dirty = False
        self._cb_queued = False

    # ---- called by the model -------------------------------------------------------------------------------
    def note_forward(self, tower: str):
        """A forward through ``tower`` whose backward will write the tower's gradient range."""
        self.expected[tower] += 1
        self.dirty = True

    def _is_last_pass(self, tower: str) -> bool:
        return self.expected[tower] > 0 and self.seen[tower] + 1 == self.expected[tower]

    def layer_cuts(self, tower: str, layers: int):
        """Blocks at which a tower's backward pauses to hand over gradients: ``pieces`` roughly equal groups.
        None for any pass but the last one through the tower (its range will be written again)."""
        if not active() or self.pieces < 2 or tower in self.done or not self._is_last_pass(tower):
            return []
        return piece_cuts(layers, self.pieces)

    def _reduce_range(self, a: int, b: int, async_op: bool):
        if b <= a:
            return
        W = world_size()
        if self.plan is not None and (a, b) != tuple(self.plan.head):
            for lo, hi in self.plan.split(a, b):
                h = _reduce_scatter_mean(self.model._flat_grad[lo:hi], W, rank(), async_op)
                if async_op and h is not None:
                    self.pending.append(h)
            return
        buf = self.model._flat_grad[a:b]
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

    def _join_fences(self, name: str):
        """Order the current stream behind the gradient writes of this tower's earlier passes."""
        if self.fences[name]:
            cur = torch.cuda.current_stream()
            for ev in self.fences[name]:
                cur.wait_event(ev)
            self.fences[name] = []

    def _on_tower(self, model, name: str, upto_layer: Optional[int] = None):
        if not active():
            return
        self.dirty = True
        if self.auto_finish and not self._cb_queued:
            # we are inside backward(): run finish() when the whole graph has been processed (DDP does the same)
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._finish_callback)
                self._cb_queued = True
            except RuntimeError:          # not inside a backward pass (direct call in a test)
                pass
        last = self._is_last_pass(name)
        if upto_layer is None:
            self.seen[name] += 1
        if name in self.done:
            return
        if not last:
            if upto_layer is None and model._flat_grad.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.fences[name].append(ev)
            return
        if model._flat_grad.is_cuda:
            self._join_fences(name)
        a, b = model._ranges[name]
        start = self.progress.get(name, a)
        end = b if upto_layer is None else model._layer_end[name][upto_layer]
        self._reduce_range(start, end, async_op=True)
        self.progress[name] = end
        if upto_layer is None:
            self.done.add(name)

    __call__ = _on_tower

    def _finish_callback(self):
        self._cb_queued = False
        self.finish()

    def finish(self, passes_per_tower: Optional[int] = None):
        """Reduce what the backward hooks have not, wait for the pending pieces, reset for the next step.
        ``passes_per_tower`` is accepted for compatibility and ignored (passes are counted, see class docstring).
        Call from the stream the step is issued on (every tower backward has made that stream wait for its
        gradient writes, functional._publish_to_main)."""
        if not active() or not self.dirty:
            self._reset()
            return
        cuda = self.model._flat_grad.is_cuda
        for name in self.TOWERS:
            if name not in self.done:
                if cuda:
                    self._join_fences(name)
                a, b = self.model._ranges[name]
                self._reduce_range(self.progress.get(name, a), b, async_op=False)
        self._reduce("head", async_op=False)
        for h in self.pending:
            h.wait()
        self.pending = []
        self._reset()


def consolidate(model, optimizer):
    """COLLECTIVE (every rank calls it): after sharded optimiser steps, gather both Adam moments from their owners so that
    ``optimizer.state_dict()`` -- the 'optimizer' entry of a checkpoint -- can be taken on any rank.  (The parameters
    themselves are whole on every rank after every step.)  No-op when nothing is stale."""
    plan = getattr(getattr(model, "grad_sync", None), "plan", None)
    if plan is None or not getattr(optimizer, "_moments_stale", False):
        return
    W, r = world_size(), rank()
    for buf in (optimizer.m, optimizer.v):
        for lo, hi in plan.pieces:
            _all_gather_in_place(buf[lo:hi], W, r)
    optimizer._moments_stale = False


class DistributedDataParallel(torch.nn.Module):
    """Call-site replacement for ``torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu],
    find_unused_parameters=True)`` (train.py:222-225).  torch's own wrapper cannot drive this model: its reducer
    waits for autograd hooks on every parameter, and the HIP backward writes parameter gradients as side effects of
    three coarse nodes, so those hooks never fire (the model raises if it finds itself inside torch's wrapper).
    This class keeps the contract the reference's loop relies on -- ``wrapped(image, text, ...)`` forwards to the
    model, ``wrapped.module`` is the model (engine.py:52-61), gradients are averaged over ranks by the time
    ``backward()`` returns -- by installing ``GradSync``.  Extra torch keyword arguments are accepted and ignored."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0, broadcast_buffers=True, process_group=None,
                 bucket_cap_mb=None, find_unused_parameters=False, sharded_optimizer: Optional[bool] = None, **ignored):
        super().__init__()
        if process_group is not None:
            raise NotImplementedError("only the default process group is supported")
        self.module = module
        self.device_ids = device_ids
        self.grad_sync = GradSync(module, sharded=sharded_optimizer)
        if is_dist() and world_size() > 1:      # DDP broadcasts rank 0's parameters at construction
            with torch.no_grad():
                for p in module.parameters():
                    dist.broadcast(p.data, src=0)
            if hasattr(module, "mark_operands_stale"):
                module.mark_operands_stale()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def consolidate(self, optimizer):
        """Collective: see ``distributed.consolidate``."""
        consolidate(self.module, optimizer)
