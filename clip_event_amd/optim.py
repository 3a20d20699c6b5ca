"""The optimiser side of the step: fused clip_grad_norm_(.,1) + Adam over the model's flat
buffers (reference engine.py:87-95, build_optimizer engine.py:129-151) -- two HIP launches per step, no host
synchronisation -- and the learning-rate schedules of utils.py:310-416 / build_lr_scheduler engine.py:154-176
(host arithmetic: one float per step)."""
from __future__ import annotations

import math
import os
from bisect import bisect_right
from ctypes import c_float, c_int, c_long
from typing import List, Sequence

import torch

from ._lib import check, lib, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, weight_decay)`` semantics (L2 decay added to the gradient)
    preceded by the global-norm clip of ``clip_grad_norm_(params, max_norm)``.  ``max_norm=None``
    disables clipping.

    A real ``torch.optim.Optimizer`` (one param group over ``model.parameters()``), so the stock and the
    reference's LR schedulers drive it through ``param_groups[0]['lr']``, and ``state_dict()`` /
    ``load_state_dict()`` speak ``torch.optim.Adam``'s format (per-parameter ``step`` / ``exp_avg`` /
    ``exp_avg_sq``): the ``'optimizer'`` entry of a reference checkpoint (engine.py:202-218) loads here and vice
    versa.  The moments themselves live in two flat buffers next to the flat parameters."""

    def __init__(self, model, lr: float = 1e-6, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_norm=1.0):
        frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
        if frozen:
            raise NotImplementedError("FusedAdam updates the whole flat parameter buffer; frozen parameters "
                                      f"({frozen[:3]}...) need torch.optim.Adam over the trainable ones instead")
        self.model = model
        self.max_norm = max_norm
        self.step_count = 0
        self.sat_poll_every = int(os.environ.get("CE_SAT_POLL_EVERY", "16"))     # 0 = never
        self.m = self.v = self.sumsq = None
        super().__init__([p for p in model.parameters() if p.requires_grad],
                         dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    # kept as attributes of the first (only) group so that schedulers and user code see one source of truth
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    @property
    def weight_decay(self):
        return self.param_groups[0]["weight_decay"]

    def _state(self):
        m = self.model
        m._ready()
        if self.m is None or self.m.numel() != m._flat.numel() or self.m.device != m._flat.device:
            self.m = torch.zeros_like(m._flat)
            self.v = torch.zeros_like(m._flat)
            self.sumsq = torch.zeros(1, dtype=torch.float32, device=m._flat.device)

    def zero_grad(self, set_to_none: bool = False):
        self._state()
        self.model._flat_grad.zero_()
        self.model._attach_grads_fast()

    def zero_grad_first_touch(self):
        """``zero_grad`` for a caller that owns the whole step (engine.train_step): the block weight gradients are left
        to be overwritten by the step's first backward pass (model.zero_grad_first_touch)."""
        self._state()
        self.model.zero_grad_first_touch()

    def grad_norm(self) -> torch.Tensor:
        """Device scalar: total L2 norm of the gradients as of the last ``step``."""
        return self.sumsq.sqrt()

    @torch.no_grad()
    def step(self, closure=None):
        """clip + Adam (engine.py:87-95): sum of squares of the whole gradient buffer, then the update with the clip coefficient
        applied on the fly -- by default in tiles that also leave the blocks' W^T operand copies behind (``ce_adam_step_tiles``)."""
        if closure is not None:
            raise RuntimeError("FusedAdam.step takes no closure")
        self._state()
        m = self.model
        m._settle_first_touch()         # a tower that saw no backward since zero_grad_first_touch
        m.wait_transposes()             # Adam rewrites the bf16 mirror an asynchronous W^T rebuild may still be reading
        n = m._flat.numel()
        s = stream()
        self.step_count += 1
        lr = float(self.param_groups[0]["lr"])
        sumsq = None
        from . import distributed as D
        plan = getattr(getattr(m, "grad_sync", None), "plan", None)
        sharded = plan is not None and D.active()
        if self.max_norm is not None:
            sumsq = self.sumsq
            if not sharded:
                self.sumsq.zero_()
                check(lib().ce_sumsq(ptr(m._flat_grad), c_long(n), ptr(self.sumsq), s), "ce_sumsq")

        def adam(lo, hi, st):
            check(lib().ce_adam_step(ptr(m._flat[lo:hi]), ptr(m._flat_grad[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                     ptr(m._flat16[lo:hi]), c_long(hi - lo), ptr(sumsq), c_float(self.max_norm or 0.0), c_float(lr),
                                     c_float(self.betas[0]), c_float(self.betas[1]), c_float(self.eps), c_float(self.weight_decay),
                                     c_int(self.step_count), st), "ce_adam_step")

        if sharded:
            # sharded step (distributed.ShardPlan; DESIGN 5 lever 2): the gradient pieces arrived reduce-SCATTERED, this rank updates
            # its shard of every piece (+ the replicated head range), the fp32 masters are completed by an all-gather in place and
            # the bf16 operand mirror is re-cast from them by the next refresh_operands
            D.sharded_update(plan, m._flat, sumsq,
                             lambda lo, hi: check(lib().ce_sumsq(ptr(m._flat_grad[lo:hi]), c_long(hi - lo), ptr(self.sumsq), s), "ce_sumsq"),
                             lambda lo, hi: adam(lo, hi, s))
            self._moments_stale = True
            m.mark_operands_stale(mirror_fresh=False)
            return
        if getattr(m, "_adam_tiles_ok", False) and os.environ.get("CE_ADAM_TILES", "1") != "0":
            # the block weights tile by tile, which also writes their W^T operand copies (no transpose pass at the start of the
            # next step); everything else by the chunk table of the first-touch zero-fill (= the complement of the block weights)
            tj, tn_, tt = m._tjobs_bwd
            seg = m._adam_segment_table()
            check(lib().ce_adam_step_tiles(ptr(m._flat), ptr(m._flat_grad), ptr(self.m), ptr(self.v), ptr(m._flat16), ptr(tj), c_int(tn_),
                                           c_int(tt), ptr(seg), c_int(seg.shape[0]), ptr(sumsq), c_float(self.max_norm or 0.0), c_float(lr),
                                           c_float(self.betas[0]), c_float(self.betas[1]), c_float(self.eps), c_float(self.weight_decay),
                                           c_int(self.step_count), s), "ce_adam_step_tiles")
            m.mark_operands_stale(mirror_fresh=True, wt_fresh=True)
        else:
            adam(0, n, s)
            m.mark_operands_stale(mirror_fresh=True)
        # fp16 streams: look at the clamp counters every few steps, without a synchronisation (the copy started by one poll is
        # examined by the next); raises model.Stream16Saturation
        if self.sat_poll_every and self.step_count % self.sat_poll_every == 0 and hasattr(m, "poll_stream16_saturation"):
            m.poll_stream16_saturation()

    # ---- torch.optim.Adam-format state (checkpoint interop) ----
    def state_dict(self):
        self._state()
        m = self.model
        if getattr(self, "_moments_stale", False):
            raise RuntimeError("the Adam moments are sharded over the ranks (sharded optimiser step): call "
                               "clip_event_amd.distributed.consolidate(model, optimizer) on EVERY rank before state_dict()")
        state = {}
        params = self.param_groups[0]["params"]
        names = {id(p): n for n, p in m.named_parameters()}
        for i, p in enumerate(params):
            o = m._offsets[names[id(p)]]
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[o:o + p.numel()].view(p.shape).clone(),
                            "exp_avg_sq": self.v[o:o + p.numel()].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        self._moments_stale = False          # (every rank loads the same tensors)
        self._state()
        m = self.model
        params = self.param_groups[0]["params"]
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(params):
            raise ValueError("optimizer state does not match: expected one group of %d parameters" % len(params))
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if k == "betas" else v
        names = {id(p): n for n, p in m.named_parameters()}
        self.m.zero_()
        self.v.zero_()
        steps = set()
        with torch.no_grad():
            for key, st in sd["state"].items():
                p = params[int(key)]
                o = m._offsets[names[id(p)]]
                self.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused kernel keeps one")
        self.step_count = steps.pop() if steps else 0


def _warmup_factor_at(method: str, it: int, warmup_iters: int, warmup_factor: float) -> float:
    """utils.py:393-416: 1 after the warm-up; before it a constant, or a line from warmup_factor to 1."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return warmup_factor
    if method == "linear":
        a = it / warmup_iters
        return warmup_factor * (1.0 - a) + a
    raise ValueError("Unknown warmup method: {}".format(method))


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    """utils.py:310-347: base_lr x warm-up x gamma^(milestones passed)."""

    def __init__(self, optimizer, milestones: Sequence[int], gamma: float = 0.1, warmup_factor: float = 0.001,
                 warmup_epochs: int = 5, warmup_method: str = "linear", last_epoch: int = -1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_epochs, self.warmup_method = warmup_factor, warmup_epochs, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self) -> List[float]:
        w = _warmup_factor_at(self.warmup_method, self.last_epoch, self.warmup_epochs, self.warmup_factor)
        k = bisect_right(self.milestones, self.last_epoch)
        return [b * w * self.gamma ** k for b in self.base_lrs]


class WarmupCosineLR(torch.optim.lr_scheduler._LRScheduler):
    """utils.py:350-390: base_lr x warm-up x half cosine over ``max_iters``."""

    def __init__(self, optimizer, max_iters: int, warmup_factor: float = 0.001, warmup_epochs: int = 5,
                 warmup_method: str = "linear", last_epoch: int = -1):
        self.max_iters = max_iters
        self.warmup_factor, self.warmup_epochs, self.warmup_method = warmup_factor, warmup_epochs, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self) -> List[float]:
        w = _warmup_factor_at(self.warmup_method, self.last_epoch, self.warmup_epochs, self.warmup_factor)
        c = 0.5 * (1.0 + math.cos(math.pi * self.last_epoch / self.max_iters))
        return [b * w * c for b in self.base_lrs]


def build_optimizer(cfg: dict, model, fused: bool = True):
    """engine.py:129-151 (``cfg['optimizer']`` in {'sgd','adam'}); 'adam' returns the fused step, which also
    performs engine.py:89's clip_grad_norm_(.,1)."""
    if cfg["optimizer"] == "sgd":
        return torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=cfg["lr"],
                               momentum=cfg["momentum"], weight_decay=cfg["weight_decay"])
    if cfg["optimizer"] == "adam":
        if fused:
            return FusedAdam(model, lr=cfg["lr"], weight_decay=cfg["weight_decay"], max_norm=1.0)
        return torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg["lr"],
                                weight_decay=cfg["weight_decay"])
    raise RuntimeError("Invalid optimizer '{}'. ".format(cfg["optimizer"]))


def build_lr_scheduler(cfg: dict, optimizer, begin_epoch: int = 0):
    """engine.py:154-176."""
    kind = cfg["lr_scheduler"]
    if kind == "multisteplr":
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=cfg["lr_steps"], gamma=cfg["lr_gamma"])
    if kind == "cosineannealinglr":
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=cfg["max_epoch"] - begin_epoch)
    if kind == "warmup":
        return WarmupCosineLR(optimizer, cfg["max_epoch"], warmup_epochs=cfg["warmup_epoch"], last_epoch=begin_epoch - 1)
    if kind == "none":
        return None
    raise RuntimeError("Invalid lr scheduler '{}'. Only MultiStepLR and CosineAnnealingLR are supported.".format(kind))
