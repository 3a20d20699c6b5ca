"""Coarse autograd nodes of the HIP path: one per tower pass, one for the logits.

Each ``forward`` issues the HIP launches through the C ABI and keeps the activation stash
(a leased workspace carved by the C++ tower runner); each ``backward`` runs the HIP backward
and accumulates parameter gradients directly into the model's flat gradient buffer (the
``nn.Parameter.grad`` views), returning ``None`` for the trigger input that only exists to put
the node on the autograd tape.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from ctypes import c_float, c_int, c_long, c_void_p

import numpy as np
import torch

from . import _lib as L
from ._lib import check, lib, ptr, stream


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def _tower_workspace(model, desc, batch: int, tag: str):
    from .model import _Lease
    nbytes = lib().ce_tower_workspace_bytes(ctypes.byref(desc), c_int(batch))
    if nbytes == 0:
        raise RuntimeError("ce_tower_workspace_bytes: " + lib().ce_last_error().decode())
    buf = model._pool.take(tag, int(nbytes), model._flat.device)
    return _Lease(model._pool, tag, buf)


def _publish_to_main(ctx):
    """Parameter gradients are written as side effects on the stream this backward ran on (the tower's side
    stream); make the stream the step was issued from wait for them, so that an optimiser / clip_grad_norm_ /
    all-reduce enqueued there after ``backward()`` sees complete gradients."""
    ms = ctx.main_stream
    if ms is not None:
        cur = torch.cuda.current_stream()
        if cur != ms:
            ms.wait_stream(cur)


def _tower_backward(model, desc, tower: str, batch: int, rows: int, cu, x0, lease, dx, sel, dx_sel):
    """Tower backward, cut into layer ranges when a data-parallel gradient exchange is attached: after every
    range the gradients of the blocks done so far are handed to ``model.grad_sync`` (they form a contiguous
    prefix of the tower's range in the flat gradient buffer, model._prepare), so their all-reduce overlaps the
    remaining blocks."""
    cl, s = lib(), stream()
    layers = desc.layers
    cuts = model.grad_sync.layer_cuts(tower, layers) if (model.grad_sync is not None and
                                                           hasattr(model.grad_sync, "layer_cuts")) else []
    model.wait_transposes()       # the blocks' W^T copies may still be in flight on the auxiliary stream (model.refresh_operands)
    # the step's first backward pass of this tower writes the block weight gradients instead of accumulating them
    # (model.zero_grad_first_touch); the flag is consumed here, so later passes of the same step accumulate
    pending = getattr(model, "_first_touch", None)
    desc.wgrad_overwrite = 1 if (pending and tower in pending) else 0
    if desc.wgrad_overwrite:
        pending.discard(tower)
    hi = layers - 1
    for lo in list(cuts) + [0]:
        check(cl.ce_tower_backward_range(ctypes.byref(desc), c_int(batch), c_int(rows), ptr(cu), ptr(x0), ptr(lease.buf),
                                         ptr(dx), ptr(sel), ptr(dx_sel), c_int(hi), c_int(lo), s),
              f"ce_tower_backward_range({tower})")
        if lo > 0:
            model.grad_sync(model, tower, upto_layer=lo)
        hi = lo - 1


class TextPacking:
    """Row layout of one caption batch in the text tower.  ``cu is None``: dense, every caption owns
    ``context_length`` rows.  Otherwise only the live tokens (SOT .. EOT) are rows: ``cu`` int32 [n+1] prefix sums,
    ``src`` int32 [rows] flat index into the [n, T] id matrix, ``sel`` int32 [n] = each caption's EOT row."""
    __slots__ = ("rows", "cu", "src", "sel", "_keep")

    def __init__(self, rows, cu, src, sel, keep=None):
        self.rows, self.cu, self.src, self.sel, self._keep = rows, cu, src, sel, keep


def attach_lengths(text: torch.Tensor, lengths) -> torch.Tensor:
    """Tag a token tensor with its captions' HOST-side lengths (tokens up to and including the EOT = ``argmax + 1`` per row,
    model_clip.py:415): the tokenizer / data loader knows them before the batch is copied to the GPU, and with them the text
    tower sizes its launches without reading anything back from the device.  The tag lives on this tensor OBJECT (a
    ``.clone()`` / ``.to()`` result needs its own)."""
    text._ce_lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64).reshape(-1))
    return text


def host_lengths(text_cpu: torch.Tensor) -> np.ndarray:
    """``argmax(dim=-1) + 1`` of a CPU token matrix (first occurrence of the row maximum, as torch / numpy both define it)."""
    return text_cpu.reshape(-1, text_cpu.shape[-1]).numpy().argmax(axis=-1).astype(np.int64) + 1


def tokens_to_device(text: torch.Tensor, device) -> torch.Tensor:
    """``text.to(device)`` of engine.py:52 for a token matrix that is still on the host: the copy is asynchronous (pinned
    staging) and the result carries the host-side lengths (``attach_lengths``)."""
    if text.is_cuda:
        return text
    lens = getattr(text, "_ce_lengths", None)
    if lens is None:
        lens = host_lengths(text)
    if torch.device(device).type == "cuda" and not text.is_pinned():
        # a copy from pageable memory blocks the host until the stream has drained (the step could not run ahead any more):
        # stage it through pinned memory ourselves (tens of KB; the caching host allocator keeps the block until the copy is done)
        pin = torch.empty(text.shape, dtype=text.dtype, pin_memory=True)
        pin.copy_(text)
        text = pin
    out = text.to(device, non_blocking=True)
    out._ce_lengths = lens
    return out


_PACK_CACHE_SLOTS = 8


def _pack_key(text):
    base = text._base if text._base is not None else text
    return (id(base), text.storage_offset(), tuple(text.shape), tuple(text.stride()), text._version), base


def text_packing(model, text, lengths=None) -> TextPacking:
    """Drop the rows after each caption's EOT (model_clip.py:415 takes the feature AT the EOT; with the causal mask of
    model_clip.py:377-384 later rows can reach neither that feature nor any gradient, so the result is unchanged).
    The lengths size the launches, so the HOST needs them: from ``lengths`` / the tensor's ``attach_lengths`` tag (what a
    tokenizer or data loader knows anyway: no device round trip), else by one 4*n-byte read-back per NEW token tensor (the
    layout of a tensor seen before -- same storage, same version -- is reused; a small cache, so the fixed entity / role
    tensors of config 4 stay cached beside the captions).  ``CE_CHECK_LENGTHS=1`` checks given lengths against the device."""
    cl, s = lib(), stream()
    n, T = text.shape
    dev = text.device
    if lengths is None:
        lengths = getattr(text, "_ce_lengths", None)
    cache = getattr(model, "_pack_cache", None)
    if not isinstance(cache, dict):
        cache = model._pack_cache = {}
    key, base = _pack_key(text)
    key = key + (bool(model.pack_text),)
    hit = cache.get(key)
    if hit is not None and hit[0]() is base:
        return hit[1]
    if lengths is not None:
        lens = np.asarray(lengths, dtype=np.int64).reshape(-1)
        if lens.shape[0] != n or lens.min() < 1 or lens.max() > T:
            raise RuntimeError(f"text lengths must be {n} values in 1..{T}")
        if os.environ.get("CE_CHECK_LENGTHS", "0") == "1":
            dev_lens = (text.argmax(dim=-1) + 1).cpu().numpy()
            if not np.array_equal(dev_lens, lens):
                raise RuntimeError("attach_lengths / text_lengths disagree with argmax(text) + 1 on the device")
        flat = np.arange(n, dtype=np.int64) * T + lens - 1
        eot = None
    else:
        eot = _empty((n,), torch.int32, dev)
        check(cl.ce_eot_rows(ptr(text), ptr(eot), c_long(n), c_int(T), s), "ce_eot_rows")
        flat = None
    if not model.pack_text:
        if eot is None:
            meta = torch.empty(n, dtype=torch.int32, pin_memory=True)
            meta.numpy()[:] = flat
            eot = meta.to(dev, non_blocking=True)
            pk = TextPacking(n * T, None, None, eot, keep=(meta,))
        else:
            pk = TextPacking(n * T, None, None, eot)
    else:
        if flat is None:
            flat = eot.cpu().numpy().astype(np.int64)                   # host sync on this stream only
        lens = flat - np.arange(n, dtype=np.int64) * T + 1
        cu = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=cu[1:])
        R = int(cu[-1])
        src = np.repeat(np.arange(n, dtype=np.int64) * T - cu[:-1], lens) + np.arange(R, dtype=np.int64)
        meta = torch.empty(2 * n + 1 + R, dtype=torch.int32, pin_memory=True)
        meta.numpy()[: n + 1] = cu
        meta.numpy()[n + 1: 2 * n + 1] = cu[1:] - 1
        meta.numpy()[2 * n + 1:] = src
        meta_d = meta.to(dev, non_blocking=True)
        pk = TextPacking(R, meta_d[: n + 1], meta_d[2 * n + 1:], meta_d[n + 1: 2 * n + 1], keep=(meta, meta_d))
    if len(cache) >= _PACK_CACHE_SLOTS:
        for k in [k for k, v in cache.items() if v[0]() is None] or [next(iter(cache))]:
            cache.pop(k, None)
    cache[key] = (weakref.ref(base), pk)
    return pk


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


def _stream_type(model):
    """(torch dtype, CE_T_* code) of the residual stream: fp32, or IEEE fp16 with ``model.stream16``."""
    if getattr(model, "stream16", False):
        return torch.float16, L.T_F16
    return torch.float32, L.T_F32


def _grad_scale(model, desc, top_grad):
    """Device scalar: the power of two this pass's fp16 gradient stream is stored multiplied by, from the largest element
    of the fp32 gradient that enters the tower (``ce_grad_scale``).  None for an fp32 stream.  Also hands it to the tower
    descriptor (``ce_tower_desc.grad_scale``)."""
    if not getattr(model, "stream16", False):
        desc.grad_scale = None
        return None
    buf = _empty((1 + 256,), torch.float32, top_grad.device)            # [0] the scale, [1:] CE_GRAD_SCALE_SCRATCH partial maxima
    check(lib().ce_grad_scale(ptr(top_grad), c_long(top_grad.numel()), c_float(model.grad_target), c_void_p(buf.data_ptr() + 4),
                              ptr(buf), stream()), "ce_grad_scale")
    desc.grad_scale = buf.data_ptr()
    return buf


def _check_stream(ctx, model):
    if ctx.stream16 != bool(getattr(model, "stream16", False)):
        raise RuntimeError("model.stream16 changed between this pass's forward and its backward")


class EncodeImageFn(torch.autograd.Function):
    """VisualTransformer.forward (model_clip.py:232-263) as one autograd node."""

    @staticmethod
    def forward(ctx, image, trigger, model, use_grid: bool):
        cl, s = lib(), stream()
        v = model.visual
        dev = model._flat.device
        if image.device != dev:
            raise RuntimeError("image must live on the model's GPU")
        image = _f32(image)                                   # image.type(self.dtype), model_clip.py:391
        B = image.shape[0]
        R, ps, g, D, E = v.input_resolution, v.patch_size, v.patch_num, model.vision_width, model.embed_dim
        if tuple(image.shape[1:]) != (3, R, R):
            raise RuntimeError(f"expected image [B,3,{R},{R}], got {tuple(image.shape)}")
        T = g * g + 1
        M = B * T
        P = model._pmap
        patches = _empty((B * g * g, model._kp), torch.bfloat16, dev)
        check(cl.ce_im2col(ptr(image), ptr(patches), c_int(B), c_int(R), c_int(ps), c_int(model._kp), s), "ce_im2col")
        patch_out = _empty((B * g * g, D), torch.float32, dev)
        wconv = model._w16["visual.conv1.weight"]
        check(cl.ce_gemm_nt(ptr(patches), c_long(model._kp), ptr(wconv), c_long(model._kp), c_int(B * g * g), c_int(D),
                            c_int(model._kp), c_int(L.EPI_F32), None, None, c_long(0), ptr(patch_out), c_long(D), None,
                            c_long(0), None, c_long(0), s), "ce_gemm_nt(conv1)")
        xpre = _empty((M, D), torch.float32, dev)
        check(cl.ce_vision_assemble(ptr(patch_out), ptr(P["visual.class_embedding"]), ptr(P["visual.positional_embedding"]),
                                    ptr(xpre), c_int(B), c_int(T), c_int(D), s), "ce_vision_assemble")
        sdt, ST = _stream_type(model)                         # residual stream: fp32, or fp16 (model.stream16)
        x0 = _empty((M, D), sdt, dev)
        mean_pre, rstd_pre = _empty((M,), torch.float32, dev), _empty((M,), torch.float32, dev)
        check(cl.ce_layernorm_fwd_t(ptr(xpre), c_int(L.T_F32), c_long(D), None, ptr(P["visual.ln_pre.weight"]),
                                    ptr(P["visual.ln_pre.bias"]), ptr(x0), c_int(ST), c_long(D), ptr(mean_pre), ptr(rstd_pre),
                                    c_int(M), c_int(D), c_float(1e-5), s), "ce_layernorm_fwd(ln_pre)")
        lease = _tower_workspace(model, model._vdesc, B, "vision")
        if use_grid:
            rows, n = None, M
        else:                      # only the CLS row of each image is consumed (model_clip.py:256): pruned last block
            cache = model.__dict__.setdefault("_cls_rows", {})
            rows = cache.get((B, T, dev))
            if rows is None:                                   # built once per batch size (two torch launches otherwise)
                rows = cache[(B, T, dev)] = (torch.arange(B, device=dev, dtype=torch.int32) * T)
            n = B
        xN = _empty((n, D), sdt, dev)
        check(cl.ce_tower_forward(ctypes.byref(model._vdesc), c_int(B), c_int(M), None, ptr(x0), ptr(lease.buf), ptr(xN),
                                  ptr(rows), s),
              "ce_tower_forward(vision)")
        hpost = _empty((n, D), torch.bfloat16, dev)
        mean_post, rstd_post = _empty((n,), torch.float32, dev), _empty((n,), torch.float32, dev)
        check(cl.ce_layernorm_fwd_t(ptr(xN), c_int(ST), c_long(D), None, ptr(P["visual.ln_post.weight"]),
                                    ptr(P["visual.ln_post.bias"]), ptr(hpost), c_int(L.T_BF16), c_long(D), ptr(mean_post),
                                    ptr(rstd_post), c_int(n), c_int(D), c_float(1e-5), s), "ce_layernorm_fwd(ln_post)")
        feat = _empty((n, E), torch.float32, dev)
        wp = model._w16t["visual.proj"]                       # [E, D]: features = hpost @ proj
        check(cl.ce_gemm_nt(ptr(hpost), c_long(D), ptr(wp), c_long(D), c_int(n), c_int(E), c_int(D), c_int(L.EPI_F32),
                            None, None, c_long(0), ptr(feat), c_long(E), None, c_long(0), None, c_long(0), s),
              "ce_gemm_nt(visual.proj)")
        ctx.model, ctx.lease, ctx.use_grid, ctx.B = model, lease, use_grid, B
        ctx.stream16 = sdt == torch.float16
        ctx.main_stream = getattr(model, "_main_stream", None)
        ctx.saved = (patches, xpre, mean_pre, rstd_pre, x0, xN, rows, hpost, mean_post, rstd_post)
        return feat.view(B, T, E) if use_grid else feat

    @staticmethod
    def backward(ctx, dfeat):
        model, lease = ctx.model, ctx.lease
        if lease.buf is None:
            raise RuntimeError("backward through encode_image a second time: the activation stash was released")
        cl, s = lib(), stream()
        patches, xpre, mean_pre, rstd_pre, x0, xN, rows, hpost, mean_post, rstd_post = ctx.saved
        v = model.visual
        dev = model._flat.device
        B, g, D, E = ctx.B, v.patch_num, model.vision_width, model.embed_dim
        T = g * g + 1
        M = B * T
        n = M if ctx.use_grid else B
        model._attach_grads()
        P, G = model._pmap, model._gview
        dfeat = _f32(dfeat).reshape(n, E)
        dfb = _empty((n, E), torch.bfloat16, dev)
        check(cl.ce_cast_bf16(ptr(dfeat), ptr(dfb), c_long(n * E), s), "ce_cast_bf16")
        # features = hpost @ proj  ->  dhpost = dF proj^T ; dproj += hpost^T dF
        dh = _empty((n, D), torch.bfloat16, dev)
        check(cl.ce_gemm_nt(ptr(dfb), c_long(E), ptr(model._w16["visual.proj"]), c_long(E), c_int(n), c_int(D), c_int(E),
                            c_int(L.EPI_BF16), None, None, c_long(0), ptr(dh), c_long(D), None, c_long(0), None,
                            c_long(0), s), "ce_gemm_nt(dproj)")
        check(cl.ce_gemm_tn(ptr(hpost), c_long(D), ptr(dfb), c_long(E), c_int(n), c_int(D), c_int(E),
                            ptr(G("visual.proj")), c_long(E), c_int(0), s), "ce_gemm_tn(visual.proj)")
        _check_stream(ctx, model)
        sdt, ST = _stream_type(model)
        # gradient w.r.t. the tower output (fp32): [B,D] in pruned mode (the tower's dx_sel) or [M,D] in grid mode, where it
        # IS the gradient stream on entry -- an fp16 stream holds gradient * scale, the scale chosen from this tensor
        grid_mode = rows is None
        dxn = _empty((n, D), torch.float32, dev)
        check(cl.ce_layernorm_bwd_t(ptr(dh), c_int(L.T_BF16), c_long(D), ptr(xN), c_int(ST), c_long(D), None, ptr(mean_post),
                                    ptr(rstd_post), ptr(P["visual.ln_post.weight"]), None, c_int(L.T_F32), ptr(dxn),
                                    c_int(L.T_F32), c_long(D), None, c_long(0), ptr(G("visual.ln_post.weight")),
                                    ptr(G("visual.ln_post.bias")), None, None, c_int(n), c_int(D), s), "ce_layernorm_bwd(ln_post)")
        gs = _grad_scale(model, model._vdesc, dxn)
        if grid_mode:
            if gs is None:
                dx = dxn
            else:
                dx = _empty((M, D), sdt, dev)
                check(cl.ce_cast_scaled(ptr(dxn), c_int(L.T_F32), ptr(dx), c_int(ST), ptr(gs), c_int(0), c_long(M * D), s),
                      "ce_cast_scaled")
            _tower_backward(model, model._vdesc, "visual", B, M, None, x0, lease, dx, None, None)
        else:
            dx = _empty((M, D), sdt, dev)
            _tower_backward(model, model._vdesc, "visual", B, M, None, x0, lease, dx, rows, dxn)
        lease.release()
        # ln_pre: x0 = LN(xpre); its dy is the gradient stream
        dxpre = _empty((M, D), torch.float32, dev)
        check(cl.ce_layernorm_bwd_t(ptr(dx), c_int(ST), c_long(D), ptr(xpre), c_int(L.T_F32), c_long(D), None, ptr(mean_pre),
                                    ptr(rstd_pre), ptr(P["visual.ln_pre.weight"]), None, c_int(L.T_F32), ptr(dxpre),
                                    c_int(L.T_F32), c_long(D), None, c_long(0), ptr(G("visual.ln_pre.weight")),
                                    ptr(G("visual.ln_pre.bias")), None, ptr(gs), c_int(M), c_int(D), s),
              "ce_layernorm_bwd(ln_pre)")
        # positional / class embedding gradients: sums over the batch axis
        check(cl.ce_batch_reduce(ptr(dxpre), ptr(G("visual.positional_embedding")), c_int(B), c_long(T * D),
                                 c_long(T * D), c_int(1), s), "ce_batch_reduce(pos)")
        check(cl.ce_batch_reduce(ptr(dxpre), ptr(G("visual.class_embedding")), c_int(B), c_long(T * D), c_long(D),
                                 c_int(1), s), "ce_batch_reduce(cls)")
        # conv1 weight gradient: dW[width, 3*p*p] += dpatch^T patches   (no input gradient is ever needed)
        dpatch = _empty((B * g * g, D), torch.bfloat16, dev)
        check(cl.ce_vision_assemble_bwd(ptr(dxpre), ptr(dpatch), c_int(B), c_int(T), c_int(D), s), "ce_vision_assemble_bwd")
        if model._conv_pad is None:
            check(cl.ce_gemm_tn(ptr(dpatch), c_long(D), ptr(patches), c_long(model._kp), c_int(B * g * g), c_int(D),
                                c_int(model._kp), ptr(G("visual.conv1.weight")), c_long(model._kp), c_int(0), s),
                  "ce_gemm_tn(conv1)")
        else:       # padded patch columns (model._build_device_tables): gradient through a padded scratch
            gp = model._conv_gpad
            gp.zero_()
            check(cl.ce_gemm_tn(ptr(dpatch), c_long(D), ptr(patches), c_long(model._kp), c_int(B * g * g), c_int(D),
                                c_int(model._kp), ptr(gp), c_long(model._kp), c_int(0), s), "ce_gemm_tn(conv1, padded)")
            check(cl.ce_add_cols(ptr(gp), c_long(model._kp), ptr(G("visual.conv1.weight")), c_long(model._kp_real), c_int(D),
                                 c_int(model._kp_real), s), "ce_add_cols(conv1)")
        if model.grad_sync is not None:
            model.grad_sync(model, "visual")
        _publish_to_main(ctx)
        return None, None, None, None


class EncodeTextFn(torch.autograd.Function):
    """CLIP.encode_text (model_clip.py:398-417) as one autograd node."""

    @staticmethod
    def forward(ctx, text, trigger, model):
        cl, s = lib(), stream()
        dev = model._flat.device
        if text.device != dev:
            raise RuntimeError("text must live on the model's GPU")
        if text.dtype != torch.int64:
            text = text.long()
        text = text.contiguous()
        n, T = text.shape
        if T != model.context_length:
            raise RuntimeError(f"expected {model.context_length} tokens per row, got {T}")
        D, E = model.transformer.width, model.embed_dim
        P = model._pmap
        pk = text_packing(model, text)
        M = pk.rows                                        # activation rows: live tokens only when packed
        sdt, ST = _stream_type(model)                      # residual stream: fp32, or fp16 (model.stream16)
        x0 = _empty((M, D), sdt, dev)
        check(cl.ce_token_embed_t(ptr(text), ptr(pk.src), ptr(P["token_embedding.weight"]), ptr(P["positional_embedding"]),
                                  ptr(x0), c_int(ST), c_long(M), c_int(T), c_int(D), c_int(model.vocab_size), s),
              "ce_token_embed")
        lease = _tower_workspace(model, model._tdesc, n, "text")
        rows = pk.sel                                      # EOT row of each caption (argmax token id, model_clip.py:415)
        xN = _empty((n, D), sdt, dev)                      # pruned last block: only the EOT rows are produced
        check(cl.ce_tower_forward(ctypes.byref(model._tdesc), c_int(n), c_int(M), ptr(pk.cu), ptr(x0), ptr(lease.buf),
                                  ptr(xN), ptr(rows), s), "ce_tower_forward(text)")
        hfin = _empty((n, D), torch.bfloat16, dev)
        mean_f, rstd_f = _empty((n,), torch.float32, dev), _empty((n,), torch.float32, dev)
        check(cl.ce_layernorm_fwd_t(ptr(xN), c_int(ST), c_long(D), None, ptr(P["ln_final.weight"]), ptr(P["ln_final.bias"]),
                                    ptr(hfin), c_int(L.T_BF16), c_long(D), ptr(mean_f), ptr(rstd_f), c_int(n), c_int(D),
                                    c_float(1e-5), s), "ce_layernorm_fwd(ln_final)")
        feat = _empty((n, E), torch.float32, dev)
        wp = model._w16t["text_projection"]                   # [E, D]
        check(cl.ce_gemm_nt(ptr(hfin), c_long(D), ptr(wp), c_long(D), c_int(n), c_int(E), c_int(D), c_int(L.EPI_F32),
                            None, None, c_long(0), ptr(feat), c_long(E), None, c_long(0), None, c_long(0), s),
              "ce_gemm_nt(text_projection)")
        ctx.model, ctx.lease, ctx.n = model, lease, n
        ctx.stream16 = sdt == torch.float16
        ctx.main_stream = getattr(model, "_main_stream", None)
        ctx.saved = (text, x0, xN, pk, hfin, mean_f, rstd_f)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model, lease, n = ctx.model, ctx.lease, ctx.n
        if lease.buf is None:
            raise RuntimeError("backward through encode_text a second time: the activation stash was released")
        cl, s = lib(), stream()
        text, x0, xN, pk, hfin, mean_f, rstd_f = ctx.saved
        dev = model._flat.device
        T, D, E = model.context_length, model.transformer.width, model.embed_dim
        M, rows = pk.rows, pk.sel
        model._attach_grads()
        P, G = model._pmap, model._gview
        dfeat = _f32(dfeat)
        dfb = _empty((n, E), torch.bfloat16, dev)
        check(cl.ce_cast_bf16(ptr(dfeat), ptr(dfb), c_long(n * E), s), "ce_cast_bf16")
        dh = _empty((n, D), torch.bfloat16, dev)
        check(cl.ce_gemm_nt(ptr(dfb), c_long(E), ptr(model._w16["text_projection"]), c_long(E), c_int(n), c_int(D),
                            c_int(E), c_int(L.EPI_BF16), None, None, c_long(0), ptr(dh), c_long(D), None, c_long(0), None,
                            c_long(0), s), "ce_gemm_nt(dtext_projection)")
        check(cl.ce_gemm_tn(ptr(hfin), c_long(D), ptr(dfb), c_long(E), c_int(n), c_int(D), c_int(E),
                            ptr(G("text_projection")), c_long(E), c_int(0), s), "ce_gemm_tn(text_projection)")
        _check_stream(ctx, model)
        sdt, ST = _stream_type(model)
        dxn = _empty((n, D), torch.float32, dev)           # [n, D] gradient at the EOT rows (dx_sel of the pruned tower: fp32)
        check(cl.ce_layernorm_bwd_t(ptr(dh), c_int(L.T_BF16), c_long(D), ptr(xN), c_int(ST), c_long(D), None, ptr(mean_f),
                                    ptr(rstd_f), ptr(P["ln_final.weight"]), None, c_int(L.T_F32), ptr(dxn), c_int(L.T_F32),
                                    c_long(D), None, c_long(0), ptr(G("ln_final.weight")), ptr(G("ln_final.bias")), None,
                                    None, c_int(n), c_int(D), s), "ce_layernorm_bwd(ln_final)")
        gs = _grad_scale(model, model._tdesc, dxn)
        dx = _empty((M, D), sdt, dev)
        _tower_backward(model, model._tdesc, "text", n, M, pk.cu, x0, lease, dx, rows, dxn)
        lease.release()
        if sdt != torch.float32:      # the embedding gradients below take the fp32 gradient in true units
            dx32 = _empty((M, D), torch.float32, dev)
            check(cl.ce_cast_scaled(ptr(dx), c_int(ST), ptr(dx32), c_int(L.T_F32), ptr(gs), c_int(1), c_long(M * D), s),
                  "ce_cast_scaled")
            dx = dx32
        if pk.cu is None:
            check(cl.ce_batch_reduce(ptr(dx), ptr(G("positional_embedding")), c_int(n), c_long(T * D), c_long(T * D),
                                     c_int(1), s), "ce_batch_reduce(text pos)")
        else:
            check(cl.ce_pos_embed_bwd_packed(ptr(dx), ptr(pk.cu), ptr(G("positional_embedding")), c_int(n), c_int(T),
                                             c_int(D), s), "ce_pos_embed_bwd_packed")
        check(cl.ce_token_embed_bwd(ptr(text), ptr(pk.src), ptr(dx), ptr(G("token_embedding.weight")), c_long(M), c_int(D),
                                    c_int(model.vocab_size), s), "ce_token_embed_bwd")
        if model.grad_sync is not None:
            model.grad_sync(model, "text")
        _publish_to_main(ctx)
        return None, None, None


def _sgemm(A, sam, sak, Bm, sbk, sbn, C, M, N, K, alpha_ptr=None, alpha=1.0, alpha_exp=0, beta=0.0):
    check(lib().ce_sgemm(ptr(A), c_long(sam), c_long(sak), ptr(Bm), c_long(sbk), c_long(sbn), ptr(C), c_long(C.stride(0)),
                         c_int(M), c_int(N), c_int(K), ptr(alpha_ptr), c_float(alpha), c_int(alpha_exp), c_float(beta),
                         stream()), "ce_sgemm")


class LogitsFn(torch.autograd.Function):
    """Feature normalisation + logits (model_clip.py:496-521).  ``text_all`` / ``image_all`` may be
    larger than the local features (all-gathered global batch): logits_per_image =
    s * I_local @ T_all^T, logits_per_text = s * T_local @ I_all^T."""

    @staticmethod
    def forward(ctx, fi, ft, logit_scale, overbatch: bool, want: str = "both"):
        cl, s = lib(), stream()
        dev = fi.device
        ctx.set_materialize_grads(False)
        fi, ft = _f32(fi), _f32(ft)
        B, E = fi.shape
        N = ft.shape[0]
        In, Tn = torch.empty_like(fi), torch.empty_like(ft)
        inv_i, inv_t = _empty((B,), torch.float32, dev), _empty((N,), torch.float32, dev)
        check(cl.ce_l2norm_fwd(ptr(fi), c_long(E), ptr(In), c_long(E), ptr(inv_i), c_int(B), c_int(E), s), "ce_l2norm_fwd")
        check(cl.ce_l2norm_fwd(ptr(ft), c_long(E), ptr(Tn), c_long(E), ptr(inv_t), c_int(N), c_int(E), s), "ce_l2norm_fwd")
        ls = logit_scale.detach().reshape(1)
        lpt = None
        # both matrices over the batch from the same features: logits_per_text IS logits_per_image^T (the same products summed in
        # the same order), so one GEMM + a transposed copy instead of two GEMMs; the backward folds the two gradients likewise
        ctx.twin = want == "both" and overbatch and os.environ.get("CE_HEAD_TWIN", "1") != "0"
        if want in ("both", "text") and not ctx.twin:
            lpt = _empty((N, B), torch.float32, dev)
            _sgemm(Tn, E, 1, In, 1, E, lpt, N, B, E, alpha_ptr=ls, alpha_exp=1)      # s * T I^T
        if want == "text":
            lpi = None
        elif overbatch:
            lpi = _empty((B, N), torch.float32, dev)
            _sgemm(In, E, 1, Tn, 1, E, lpi, B, N, E, alpha_ptr=ls, alpha_exp=1)      # s * I T^T
            if ctx.twin:
                lpt = lpi.t().contiguous()
        else:
            if N % B != 0:
                raise RuntimeError("per-instance logits need the same number of descriptions per image")
            K = N // B
            lpi = _empty((B, K), torch.float32, dev)
            check(cl.ce_instance_logits(ptr(In), ptr(Tn), ptr(ls), ptr(lpi), c_int(B), c_int(K), c_int(E), s),
                  "ce_instance_logits")
        # the logits are this node's OUTPUTS: keeping the tensors themselves on ctx would close a cycle through their
        # grad_fn (= ctx) that Python's collector cannot see, and every step's whole graph (both towers' activation
        # stashes) would stay alive; detached aliases share the storage without the back edge
        ctx.saved = (In, Tn, inv_i, inv_t, ls, None if lpi is None else lpi.detach(), None if lpt is None else lpt.detach())
        ctx.overbatch = overbatch
        return lpi, lpt

    @staticmethod
    def backward(ctx, dlpi, dlpt):
        cl, s = lib(), stream()
        In, Tn, inv_i, inv_t, ls, lpi, lpt = ctx.saved
        dev = In.device
        B, E = In.shape
        N = Tn.shape[0]
        twin = ctx.twin and dlpi is not None and dlpt is not None
        dIn = torch.empty_like(In) if twin else torch.zeros_like(In)      # (twin: written, not accumulated)
        dTn = torch.empty_like(Tn) if twin else torch.zeros_like(Tn)
        dls = torch.zeros(1, dtype=torch.float32, device=dev)
        if twin:
            # lpt = lpi^T: dI = s (dlpi + dlpt^T) T, dT = s (dlpi + dlpt^T)^T I, ds = <dlpi + dlpt^T, lpi> / s-scaled as below
            G = torch.add(_f32(dlpi), _f32(dlpt).t())
            _sgemm(G, N, 1, Tn, E, 1, dIn, B, E, N, alpha_ptr=ls, alpha_exp=1)
            _sgemm(G, 1, N, In, E, 1, dTn, N, E, B, alpha_ptr=ls, alpha_exp=1)
            check(cl.ce_dot(ptr(G), ptr(lpi), c_long(G.numel()), ptr(dls), s), "ce_dot")
            dlpi = dlpt = None
        if dlpt is not None:
            dlpt = _f32(dlpt)
            # lpt = s T I^T : dT += s dlpt I ; dI += s dlpt^T T
            _sgemm(dlpt, B, 1, In, E, 1, dTn, N, E, B, alpha_ptr=ls, alpha_exp=1, beta=1.0)
            _sgemm(dlpt, 1, B, Tn, E, 1, dIn, B, E, N, alpha_ptr=ls, alpha_exp=1, beta=1.0)
            check(cl.ce_dot(ptr(dlpt), ptr(lpt), c_long(N * B), ptr(dls), s), "ce_dot")
        if dlpi is not None:
            dlpi = _f32(dlpi)
            if ctx.overbatch:
                _sgemm(dlpi, N, 1, Tn, E, 1, dIn, B, E, N, alpha_ptr=ls, alpha_exp=1, beta=1.0)
                _sgemm(dlpi, 1, N, In, E, 1, dTn, N, E, B, alpha_ptr=ls, alpha_exp=1, beta=1.0)
            else:
                K = N // B
                check(cl.ce_instance_logits_bwd(ptr(dlpi), ptr(In), ptr(Tn), ptr(ls), ptr(dIn), ptr(dTn), c_int(B),
                                                c_int(K), c_int(E), s), "ce_instance_logits_bwd")
            check(cl.ce_dot(ptr(dlpi), ptr(lpi), c_long(dlpi.numel()), ptr(dls), s), "ce_dot")
        dfi, dft = torch.empty_like(In), torch.empty_like(Tn)
        check(cl.ce_l2norm_bwd(ptr(dIn), c_long(E), ptr(In), c_long(E), ptr(inv_i), ptr(dfi), c_long(E), c_int(B), c_int(E),
                               c_int(0), s), "ce_l2norm_bwd")
        check(cl.ce_l2norm_bwd(ptr(dTn), c_long(E), ptr(Tn), c_long(E), ptr(inv_t), ptr(dft), c_long(E), c_int(N), c_int(E),
                               c_int(0), s), "ce_l2norm_bwd")
        return dfi, dft, dls.reshape(()), None, None


def logits_from_features(image_features, text_features, logit_scale, overbatch: bool = True, want: str = "both"):
    """``want`` in {"both", "image", "text"}: skip the logits matrix that is not needed (global-batch
    path: each rank needs only its own row blocks)."""
    return LogitsFn.apply(image_features, text_features, logit_scale, bool(overbatch), want)


class InfoNCEFn(torch.autograd.Function):
    """mean_r CE(s q^_r . k^_c, label_r) over the query rows ``sel`` (index_pos) of ``q`` against all rows of ``k`` --
    model_clip.py:496-521 + :633-662 for one direction -- without the [nq, nk] logits matrix (``ce_infonce_fwd/bwd``:
    similarity tiles on the fp32 matrix instruction, online log-sum-exp, backward from the saved log-sum-exp)."""

    @staticmethod
    def forward(ctx, q, k, logit_scale, labels, sel):
        cl, s = lib(), stream()
        dev = q.device
        q, k = _f32(q), _f32(k)
        E = q.shape[1]
        qn, kn = torch.empty_like(q), torch.empty_like(k)
        inv_q, inv_k = _empty((q.shape[0],), torch.float32, dev), _empty((k.shape[0],), torch.float32, dev)
        check(cl.ce_l2norm_fwd(ptr(q), c_long(E), ptr(qn), c_long(E), ptr(inv_q), c_int(q.shape[0]), c_int(E), s), "ce_l2norm_fwd")
        check(cl.ce_l2norm_fwd(ptr(k), c_long(E), ptr(kn), c_long(E), ptr(inv_k), c_int(k.shape[0]), c_int(E), s), "ce_l2norm_fwd")
        labels = labels.to(device=dev, dtype=torch.int64).contiguous()
        sel = sel.to(device=dev, dtype=torch.int64).contiguous() if sel is not None else None
        nq = q.shape[0] if sel is None else sel.shape[0]
        ls = logit_scale.detach().reshape(1)
        lse = _empty((nq,), torch.float32, dev)
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        cl.ce_infonce_workspace_bytes.restype = ctypes.c_size_t
        ws = _empty((int(cl.ce_infonce_workspace_bytes(c_int(nq))),), torch.uint8, dev)
        check(cl.ce_infonce_fwd(ptr(qn), c_long(E), ptr(sel), c_int(nq), ptr(kn), c_long(E), c_int(k.shape[0]), c_int(E),
                                ptr(ls), ptr(labels), ptr(lse), ptr(loss), ptr(ws), s), "ce_infonce_fwd")
        ctx.saved = (qn, kn, inv_q, inv_k, ls, labels, sel, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        cl, s = lib(), stream()
        qn, kn, inv_q, inv_k, ls, labels, sel, lse = ctx.saved
        E = qn.shape[1]
        nq = qn.shape[0] if sel is None else sel.shape[0]
        g = g.contiguous().float().reshape(1)
        dqn, dkn = torch.zeros_like(qn), torch.zeros_like(kn)
        dls = torch.zeros(1, dtype=torch.float32, device=qn.device)
        check(cl.ce_infonce_bwd(ptr(qn), c_long(E), ptr(sel), c_int(nq), ptr(kn), c_long(E), c_int(kn.shape[0]), c_int(E),
                                ptr(ls), ptr(labels), ptr(lse), ptr(g), ptr(dqn), ptr(dkn), ptr(dls), s), "ce_infonce_bwd")
        dq, dk = torch.empty_like(qn), torch.empty_like(kn)
        check(cl.ce_l2norm_bwd(ptr(dqn), c_long(E), ptr(qn), c_long(E), ptr(inv_q), ptr(dq), c_long(E), c_int(qn.shape[0]),
                               c_int(E), c_int(0), s), "ce_l2norm_bwd")
        check(cl.ce_l2norm_bwd(ptr(dkn), c_long(E), ptr(kn), c_long(E), ptr(inv_k), ptr(dk), c_long(E), c_int(kn.shape[0]),
                               c_int(E), c_int(0), s), "ce_l2norm_bwd")
        return dq, dk, dls.reshape(()), None, None


def fused_head_ok(embed_dim: int) -> bool:
    return embed_dim % 128 == 0 and 128 <= embed_dim <= 1024


class InfoNCEPairFn(torch.autograd.Function):
    """Both directions of the batch criterion on ONE pair of feature matrices (single process: the keys of one direction are
    the queries of the other): loss_i = CE(s I^ T^T, labels_i), loss_t = CE over the ``sel`` rows of s T^ I^T.  Same kernels
    and the same arithmetic as two ``InfoNCEFn`` nodes, but each matrix is normalised once, both backward directions
    accumulate into one pair of gradient buffers (``ce_infonce_bwd`` adds), and one ``ce_l2norm_bwd`` per matrix follows:
    14 launches instead of 26 on the stretch between the towers' forward and backward, where nothing else can run."""

    @staticmethod
    def forward(ctx, fi, ft, logit_scale, labels_i, labels_t, sel):
        cl, s = lib(), stream()
        dev = fi.device
        fi, ft = _f32(fi), _f32(ft)
        E = fi.shape[1]
        In, Tn = torch.empty_like(fi), torch.empty_like(ft)
        inv_i, inv_t = _empty((fi.shape[0],), torch.float32, dev), _empty((ft.shape[0],), torch.float32, dev)
        check(cl.ce_l2norm_fwd(ptr(fi), c_long(E), ptr(In), c_long(E), ptr(inv_i), c_int(fi.shape[0]), c_int(E), s), "ce_l2norm_fwd")
        check(cl.ce_l2norm_fwd(ptr(ft), c_long(E), ptr(Tn), c_long(E), ptr(inv_t), c_int(ft.shape[0]), c_int(E), s), "ce_l2norm_fwd")
        labels_i = labels_i.to(device=dev, dtype=torch.int64).contiguous()
        labels_t = labels_t.to(device=dev, dtype=torch.int64).contiguous()
        sel = sel.to(device=dev, dtype=torch.int64).contiguous() if sel is not None else None
        nqi = fi.shape[0]
        nqt = ft.shape[0] if sel is None else sel.shape[0]
        ls = logit_scale.detach().reshape(1)
        lse_i, lse_t = _empty((nqi,), torch.float32, dev), _empty((nqt,), torch.float32, dev)
        losses = torch.zeros(2, dtype=torch.float32, device=dev)
        cl.ce_infonce_workspace_bytes.restype = ctypes.c_size_t
        ws = _empty((int(cl.ce_infonce_workspace_bytes(c_int(max(nqi, nqt)))),), torch.uint8, dev)
        check(cl.ce_infonce_fwd(ptr(In), c_long(E), None, c_int(nqi), ptr(Tn), c_long(E), c_int(ft.shape[0]), c_int(E), ptr(ls),
                                ptr(labels_i), ptr(lse_i), ptr(losses[0:1]), ptr(ws), s), "ce_infonce_fwd(image)")
        check(cl.ce_infonce_fwd(ptr(Tn), c_long(E), ptr(sel), c_int(nqt), ptr(In), c_long(E), c_int(fi.shape[0]), c_int(E), ptr(ls),
                                ptr(labels_t), ptr(lse_t), ptr(losses[1:2]), ptr(ws), s), "ce_infonce_fwd(text)")
        ctx.saved = (In, Tn, inv_i, inv_t, ls, labels_i, labels_t, sel, lse_i, lse_t)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_i, g_t):
        cl, s = lib(), stream()
        In, Tn, inv_i, inv_t, ls, labels_i, labels_t, sel, lse_i, lse_t = ctx.saved
        dev = In.device
        E = In.shape[1]
        nqi = In.shape[0]
        nqt = Tn.shape[0] if sel is None else sel.shape[0]
        zero = torch.zeros(2, dtype=torch.float32, device=dev)
        g = torch.stack([zero[0] if g_i is None else g_i.float().reshape(()), zero[1] if g_t is None else g_t.float().reshape(())])
        acc = torch.zeros(In.numel() + Tn.numel() + 4, dtype=torch.float32, device=dev)         # dIn | dTn | dlogit_scale: one fill
        dIn, dTn, dls = acc[:In.numel()].view_as(In), acc[In.numel():In.numel() + Tn.numel()].view_as(Tn), acc[In.numel() + Tn.numel():]
        check(cl.ce_infonce_bwd(ptr(In), c_long(E), None, c_int(nqi), ptr(Tn), c_long(E), c_int(Tn.shape[0]), c_int(E), ptr(ls),
                                ptr(labels_i), ptr(lse_i), ptr(g[0:1]), ptr(dIn), ptr(dTn), ptr(dls), s), "ce_infonce_bwd(image)")
        check(cl.ce_infonce_bwd(ptr(Tn), c_long(E), ptr(sel), c_int(nqt), ptr(In), c_long(E), c_int(In.shape[0]), c_int(E), ptr(ls),
                                ptr(labels_t), ptr(lse_t), ptr(g[1:2]), ptr(dTn), ptr(dIn), ptr(dls), s), "ce_infonce_bwd(text)")
        dfi, dft = torch.empty_like(In), torch.empty_like(Tn)
        check(cl.ce_l2norm_bwd(ptr(dIn), c_long(E), ptr(In), c_long(E), ptr(inv_i), ptr(dfi), c_long(E), c_int(In.shape[0]), c_int(E),
                               c_int(0), s), "ce_l2norm_bwd")
        check(cl.ce_l2norm_bwd(ptr(dTn), c_long(E), ptr(Tn), c_long(E), ptr(inv_t), ptr(dft), c_long(E), c_int(Tn.shape[0]), c_int(E),
                               c_int(0), s), "ce_l2norm_bwd")
        return dfi, dft, dls[0].reshape(()), None, None, None


def small_head_ok(fi, ft, index_pos) -> bool:
    """Shapes the three-launch head (csrc/head_small.hip) takes."""
    nI, E = fi.shape
    nT = ft.shape[0]
    nsel = nT if index_pos is None else int(index_pos.shape[0])
    return (fi.dim() == 2 and ft.dim() == 2 and ft.shape[1] == E and 1 <= nI <= 1024 and 1 <= nT <= 1024 and 1 <= nsel <= nT
            and E % 4 == 0 and E <= 1024)


class SmallHeadFn(torch.autograd.Function):
    """Normalisation + logits over the batch + CriterionContrastive('ce') (model_clip.py:496-521, :633-662) as ONE node of three
    launches -- two in the forward, one in the backward -- for batches whose logits matrix is small (config 2: 256 x 256):
    between the towers' forward and backward nothing else runs, and the general head is 21 launches / 200 us there."""

    @staticmethod
    def forward(ctx, fi, ft, logit_scale, labels_i, labels_t, sel):
        cl, s = lib(), stream()
        dev = fi.device
        fi, ft = _f32(fi), _f32(ft)
        nI, E = fi.shape
        nT = ft.shape[0]
        labels_i = labels_i.to(device=dev, dtype=torch.int64).contiguous()
        labels_t = labels_t.to(device=dev, dtype=torch.int64).contiguous()
        sel = sel.to(device=dev, dtype=torch.int64).contiguous() if sel is not None else None
        nsel = nT if sel is None else int(sel.shape[0])
        cl.ce_head_small_workspace_floats.restype = ctypes.c_size_t
        cl.ce_head_small_scalars_offset.restype = ctypes.c_size_t
        dims = (c_int(nI), c_int(nT), c_int(nsel), c_int(E))
        ws = _empty((int(cl.ce_head_small_workspace_floats(*dims)),), torch.float32, dev)
        off = int(cl.ce_head_small_scalars_offset(*dims))
        ls = logit_scale.detach().reshape(1)
        check(cl.ce_head_small_fwd(ptr(fi), ptr(ft), c_int(nI), c_int(nT), c_int(E), ptr(ls), ptr(labels_i), ptr(labels_t), ptr(sel),
                                   c_int(nsel), ptr(ws), s), "ce_head_small_fwd")
        ctx.saved = (ws, ls, sel, (nI, nT, nsel, E))
        return ws[off], ws[off + 1]

    @staticmethod
    def backward(ctx, g_i, g_t):
        cl, s = lib(), stream()
        ws, ls, sel, (nI, nT, nsel, E) = ctx.saved
        dev = ws.device
        g_i = None if g_i is None else g_i.float()
        g_t = None if g_t is None else g_t.float()
        dfi, dft = _empty((nI, E), torch.float32, dev), _empty((nT, E), torch.float32, dev)
        dls = _empty((1,), torch.float32, dev)
        check(cl.ce_head_small_bwd(c_int(nI), c_int(nT), c_int(nsel), c_int(E), ptr(ls), ptr(g_i), ptr(g_t), ptr(sel), ptr(ws),
                                   ptr(dfi), ptr(dft), ptr(dls), s), "ce_head_small_bwd")
        return dfi, dft, dls.reshape(()), None, None, None


def small_contrastive_losses(fi, ft, logit_scale, labels_per_image, labels_per_text, index_pos):
    loss_i, loss_t = SmallHeadFn.apply(fi, ft, logit_scale, labels_per_image, labels_per_text, index_pos)
    return {"loss_i": loss_i, "loss_t": loss_t}


def fused_contrastive_losses(fi, ft, fi_all, ft_all, logit_scale, labels_per_image, labels_per_text, index_pos):
    """``CriterionContrastive('ce')`` over the batch on raw features: loss_i = CE(s I^ T_all^T, labels_per_image),
    loss_t = CE over the ``index_pos`` rows of s T^ I_all^T (model_clip.py:633-662), the logits never materialised."""
    if fi_all is fi and ft_all is ft:          # one process: one node for both directions
        loss_i, loss_t = InfoNCEPairFn.apply(fi, ft, logit_scale, labels_per_image, labels_per_text, index_pos)
        return {"loss_i": loss_i, "loss_t": loss_t}
    return {"loss_i": InfoNCEFn.apply(fi, ft_all, logit_scale, labels_per_image, None),
            "loss_t": InfoNCEFn.apply(ft, fi_all, logit_scale, labels_per_text, index_pos)}
