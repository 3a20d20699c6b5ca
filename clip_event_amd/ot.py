"""IPOT optimal-transport distance on the HIP kernel (reference model_ot.py:66-83)."""
from __future__ import annotations

from ctypes import c_float, c_int, c_long

import torch

from ._lib import check, lib, ptr, stream


def _rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 with unit column stride (the kernels take batch / row strides, so views such as
    ``object_vec[:, 1:]`` are read in place)."""
    t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


class _OTDistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, txt, img, txt_pad, img_pad, beta, iters):
        if not txt.is_cuda:
            raise RuntimeError("clip_event_amd OT runs on the GPU only (no CPU fallback)")
        txt, img = _rows(txt), _rows(img)
        B, M, D = txt.shape
        N = img.shape[1]
        dev = txt.device
        tp = txt_pad.to(device=dev, dtype=torch.uint8).contiguous()
        ip = img_pad.to(device=dev, dtype=torch.uint8).contiguous()
        dist = torch.empty(B, dtype=torch.float32, device=dev)
        T = torch.empty(B, N, M, dtype=torch.float32, device=dev)
        xinv = torch.empty(B, M, dtype=torch.float32, device=dev)
        yinv = torch.empty(B, N, dtype=torch.float32, device=dev)
        check(lib().ce_ot_fwd(ptr(txt), c_long(txt.stride(0)), c_long(txt.stride(1)), ptr(img), c_long(img.stride(0)),
                              c_long(img.stride(1)), ptr(tp), ptr(ip), ptr(dist), ptr(T), ptr(xinv), ptr(yinv), c_int(B),
                              c_int(M), c_int(N), c_int(D), c_float(beta), c_int(iters), stream()), "ce_ot_fwd")
        ctx.saved = (txt, img, T, xinv, yinv)
        return dist

    @staticmethod
    def backward(ctx, g):
        txt, img, T, xinv, yinv = ctx.saved
        B, M, D = txt.shape
        N = img.shape[1]
        g = g.contiguous().float()
        dtxt = torch.empty(B, M, D, dtype=torch.float32, device=txt.device)
        dimg = torch.empty(B, N, D, dtype=torch.float32, device=txt.device)
        check(lib().ce_ot_bwd(ptr(txt), c_long(txt.stride(0)), c_long(txt.stride(1)), ptr(img), c_long(img.stride(0)),
                              c_long(img.stride(1)), ptr(T), ptr(xinv), ptr(yinv), ptr(g), ptr(dtxt), ptr(dimg), c_int(B),
                              c_int(M), c_int(N), c_int(D), stream()), "ce_ot_bwd")
        return dtxt, dimg, None, None, None, None


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, cost=None, beta=0.5, iteration=50, k=1):
    """[B,M,D], [B,N,D], [B,M] bool, [B,N] bool -> [B] (model_ot.py:66-83).  ``cost`` / ``k`` other than
    the reference's call-site values (None / 1) are not supported."""
    if cost is not None or k != 1:
        raise NotImplementedError("only the call form used by CriterionAlignment is implemented (cost=None, k=1)")
    return _OTDistFn.apply(txt_emb, img_emb, txt_pad, img_pad, float(beta), int(iteration))
