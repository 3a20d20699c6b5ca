"""Drop-ins for ``CriterionContrastive`` / ``CriterionAlignment`` (reference model_clip.py:620-715)
on the HIP kernels of head.hip / ot.hip."""
from __future__ import annotations

from ctypes import c_int, c_long

import torch
from torch import nn

from ._lib import check, lib, ptr, stream


class _CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) over the rows selected by ``sel`` (index_pos) or all rows."""

    @staticmethod
    def forward(ctx, logits, labels, sel):
        if not logits.is_cuda:
            raise RuntimeError("clip_event_amd losses run on the GPU only (no CPU fallback)")
        logits = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.contiguous().float()
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        if sel is not None:
            sel = sel.to(device=logits.device, dtype=torch.int64).contiguous()
        nrows = logits.shape[0] if sel is None else sel.shape[0]
        C = logits.shape[1]
        lse = torch.empty(nrows, dtype=torch.float32, device=logits.device)
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        check(lib().ce_xent_fwd(ptr(logits), c_long(logits.stride(0)), ptr(labels), ptr(sel), ptr(lse), ptr(loss),
                                c_int(nrows), c_int(C), stream()), "ce_xent_fwd")
        ctx.saved = (logits, labels, sel, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, sel, lse = ctx.saved
        nrows = logits.shape[0] if sel is None else sel.shape[0]
        d = torch.zeros_like(logits) if sel is not None else torch.empty_like(logits)
        g = g.contiguous().float()
        check(lib().ce_xent_bwd(ptr(logits), c_long(logits.stride(0)), ptr(labels), ptr(sel), ptr(lse), ptr(g), ptr(d),
                                c_long(d.stride(0)), c_int(nrows), c_int(logits.shape[1]), stream()), "ce_xent_bwd")
        return d, None, None


def cross_entropy(logits, labels, sel=None):
    return _CrossEntropyFn.apply(logits, labels, sel)


class _ElemLossFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss() (mode 0) / nn.KLDivLoss() (mode 1), default 'mean' reduction."""

    @staticmethod
    def forward(ctx, x, y, mode):
        x = x.contiguous().float()
        y = y.to(x.device).contiguous().float()
        if x.shape != y.shape:
            raise RuntimeError(f"target size {tuple(y.shape)} must match input size {tuple(x.shape)}")
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        check(lib().ce_elem_loss_fwd(ptr(x), ptr(y), c_long(x.numel()), c_int(mode), ptr(loss), stream()), "ce_elem_loss_fwd")
        ctx.saved, ctx.mode = (x, y), mode
        return loss

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved
        dx = torch.empty_like(x)
        g = g.contiguous().float()
        check(lib().ce_elem_loss_bwd(ptr(x), ptr(y), c_long(x.numel()), c_int(ctx.mode), ptr(g), ptr(dx), stream()),
              "ce_elem_loss_bwd")
        return dx, None, None


class CriterionContrastive(nn.Module):
    """model_clip.py:620-662.  ``forward`` keeps the reference signature; ``index_pos`` is
    effectively required there (``index_select(index=None)`` fails) and is required here too."""

    def __init__(self, constrastive_loss):
        super().__init__()
        if constrastive_loss not in ("ce", "bce", "kl"):
            raise RuntimeError("Invalid constrastive_loss '{}'. ".format(constrastive_loss))
        self.kind = constrastive_loss

    def forward(self, logits_per_image, logits_per_text, labels_per_image=None, labels_per_text=None,
                index_pos=None, constrastive_overbatch=True):
        n = logits_per_image.shape[0]
        dev = logits_per_image.device
        if labels_per_image is None:
            labels_per_image = torch.arange(n, device=dev)
        if labels_per_text is None:
            labels_per_text = torch.arange(n, device=dev)
        if index_pos is None:
            raise TypeError("index_select(): argument 'index' must be Tensor, not NoneType")
        if self.kind == "ce":
            loss_i = cross_entropy(logits_per_image, labels_per_image)
        elif self.kind == "bce":
            loss_i = _ElemLossFn.apply(logits_per_image, labels_per_image, 0)
        else:
            loss_i = _ElemLossFn.apply(logits_per_image, labels_per_image, 1)
        loss_t = cross_entropy(logits_per_text, labels_per_text, index_pos)
        return {"loss_i": loss_i, "loss_t": loss_t}


class CriterionAlignment(nn.Module):
    """model_clip.py:664-715: IPOT optimal-transport distance between entity-text and object
    features (object slot 0 = whole image is dropped), summed over the batch, times 0.01."""

    def forward(self, entitytxt_vec, object_vec, entitytxt_num, object_num):
        from .ot import optimal_transport_dist
        img = object_vec[:, 1:]
        txt_pad = entitytxt_num == 0
        img_pad = object_num[:, 1:] == 0
        ot_dist = optimal_transport_dist(entitytxt_vec, img, txt_pad, img_pad).to(entitytxt_vec.dtype)
        return {"loss_ot": ot_dist.sum() * 0.01}
