"""Fused clip_grad_norm_(.,1) + Adam over the model's flat buffers (reference engine.py:87-95,
build_optimizer engine.py:129-151).  Two HIP launches per step, no host synchronisation."""
from __future__ import annotations

from ctypes import c_float, c_int, c_long

import torch

from ._lib import check, lib, ptr, stream


class FusedAdam:
    """``torch.optim.Adam(params, lr, weight_decay)`` semantics (L2 decay added to the gradient)
    preceded by the global-norm clip of ``clip_grad_norm_(params, max_norm)``.  ``max_norm=None``
    disables clipping."""

    def __init__(self, model, lr: float = 1e-6, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_norm=1.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.step_count = 0
        self.m = self.v = self.sumsq = None
        self.param_groups = [{"lr": lr}]

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

    def grad_norm(self) -> torch.Tensor:
        """Device scalar: total L2 norm of the gradients as of the last ``step``."""
        return self.sumsq.sqrt()

    def step(self):
        self._state()
        m = self.model
        n = m._flat.numel()
        s = stream()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        sumsq = None
        if self.max_norm is not None:
            self.sumsq.zero_()
            check(lib().ce_sumsq(ptr(m._flat_grad), c_long(n), ptr(self.sumsq), s), "ce_sumsq")
            sumsq = self.sumsq
        check(lib().ce_adam_step(ptr(m._flat), ptr(m._flat_grad), ptr(self.m), ptr(self.v), ptr(m._flat16), c_long(n), ptr(sumsq),
                                 c_float(self.max_norm or 0.0), c_float(lr), c_float(self.betas[0]), c_float(self.betas[1]),
                                 c_float(self.eps), c_float(self.weight_decay), c_int(self.step_count), s), "ce_adam_step")
        m.mark_operands_stale(mirror_fresh=True)
