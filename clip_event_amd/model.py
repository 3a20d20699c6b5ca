"""Drop-in for the reference's ``model_clip.CLIP`` / ``build_model`` on MI355X.

Host-side mirror of ``/root/reference/src/clip-event/model_clip.py`` (ViT towers only):
same constructor signature (incl. the ``constrastive_overbatch`` spelling), attribute names,
state-dict keys/shapes, ``forward`` / ``encode_image`` / ``encode_text`` / ``sim_entity`` /
``set_hyps`` behaviour and return arity.  All device math is hand-written HIP behind the C ABI
(``include/clip_event_hip.h``); PyTorch only owns memory, the autograd graph edges between
the coarse ops and the process group.  No CPU / eager fallback: without the HIP library every
compute call raises.

Memory layout (MI355X-first): all fp32 master parameters live in ONE flat HBM buffer and all
gradients in another (``nn.Parameter``s are views), so the optimiser is two launches and the
data-parallel gradient exchange is one bucket per tower; bf16 GEMM operand copies (straight
and transposed) are refreshed from the masters when their version counters move.
"""
from __future__ import annotations

import ctypes
import math
import os
from ctypes import c_float, c_int, c_long, c_void_p
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib as L
from ._lib import check, lib, ptr, stream
from .utils_image import patch_from_norm_bbox


# --------------------------------------------------------------------------- ctypes structs

class _BlockParams(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "ln1_w", "ln1_b", "ln2_w", "ln2_b", "b_qkv", "b_out", "b_fc", "b_proj",
        "w_qkv", "w_out", "w_fc", "w_proj", "wt_qkv", "wt_out", "wt_fc", "wt_proj",
        "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b", "g_b_qkv", "g_b_out", "g_b_fc", "g_b_proj",
        "g_w_qkv", "g_w_out", "g_w_fc", "g_w_proj",
        "w8_qkv", "w8_out", "w8_fc", "w8_proj", "s8_qkv", "s8_out", "s8_fc", "s8_proj",
        "wt8_qkv", "wt8_out", "wt8_fc", "wt8_proj", "st8_qkv", "st8_out", "st8_fc", "st8_proj")]


class _TowerDesc(ctypes.Structure):
    _fields_ = [("layers", c_int), ("width", c_int), ("heads", c_int), ("tokens", c_int), ("causal", c_int),
                ("blocks", ctypes.POINTER(_BlockParams)), ("fp8", c_int), ("stream16", c_int), ("wgrad_overwrite", c_int), ("grad_scale", c_void_p)]


# --------------------------------------------------------------------------- parameter holders
# Plain containers that reproduce the reference's module tree so that state_dict() yields the
# same keys in the same order (model_clip.py:171-183, :214-230, :317-330).

class _Affine(nn.Module):
    def __init__(self, d: int, bias_init: float = 0.0):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Linear(nn.Module):
    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        bound = 1.0 / math.sqrt(d_in)
        self.weight = nn.Parameter(torch.empty(d_out, d_in).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(d_out).uniform_(-bound, bound))


class _MHAParams(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _Linear(d, d)
        with torch.no_grad():
            self.out_proj.bias.zero_()


class _MLP(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.c_fc = _Linear(d, 4 * d)
        self.c_proj = _Linear(4 * d, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = _MHAParams(d_model)
        self.ln_1 = _Affine(d_model)
        self.mlp = _MLP(d_model)
        self.ln_2 = _Affine(d_model)
        self.n_head = n_head


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])


class _Conv1(nn.Module):
    def __init__(self, width: int, patch: int):
        super().__init__()
        fan_in = 3 * patch * patch
        bound = 1.0 / math.sqrt(fan_in)
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch).uniform_(-bound, bound))


class VisualTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.patch_num = input_resolution // patch_size
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.patch_num ** 2 + 1, width))
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.conv1 = _Conv1(width, patch_size)
        self.ln_pre = _Affine(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = _Affine(width)


class _Embedding(nn.Module):
    def __init__(self, vocab: int, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(vocab, d))


# --------------------------------------------------------------------------- workspace pool

class _Pool:
    """Reusable device buffers keyed by (tag, bytes); a buffer is busy from a forward until the
    matching backward (or until the autograd context is dropped)."""

    def __init__(self):
        self.free: Dict[Tuple, List[torch.Tensor]] = {}

    def take(self, tag, nbytes: int, device) -> torch.Tensor:
        key = (tag, nbytes, str(device))
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def give(self, tag, buf: torch.Tensor):
        self.free.setdefault((tag, buf.numel(), str(buf.device)), []).append(buf)


class _Lease:
    """Returns its buffer to the pool when garbage-collected (ctx freed) or released."""

    def __init__(self, pool: _Pool, tag, buf):
        self.pool, self.tag, self.buf = pool, tag, buf

    def release(self):
        if self.buf is not None:
            self.pool.give(self.tag, self.buf)
            self.buf = None

    def __del__(self):
        self.release()


class Stream16Saturation(FloatingPointError):
    """An fp16 residual-stream activation or gradient-stream value hit the +-65504 clamp (``CLIP.stream16``)."""

    def __init__(self, forward: int, gradient: int):
        self.forward, self.gradient = forward, gradient
        super().__init__(
            f"fp16 stream saturated: {forward} forward-stream and {gradient} gradient-stream LayerNorm slots saw a value at the "
            "+-65504 clamp; the affected step(s) used a clipped activation / gradient.  Set model.stream16 = False "
            "(CE_STREAM16=0: fp32 streams, as the reference) or, for the gradient stream, lower CE_GRAD_TARGET.")


# --------------------------------------------------------------------------- the model

_GEMM_SUFFIXES = ("attn.in_proj_weight", "attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")


class CLIP(nn.Module):
    """``CLIP`` of model_clip.py:266-552 (ViT vision tower)."""

    def __init__(self, embed_dim: int, image_resolution: int, vision_layers: int, vision_width: int,
                 vision_patch_size: int, context_length: int, vocab_size: int, transformer_width: int,
                 transformer_heads: int, transformer_layers: int, constrastive_overbatch=True, alignment=False,
                 multiattention=False):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet towers are out of scope (BASELINE.json names ViT only)")
        self.context_length = context_length
        self.vision_width = vision_width
        self.embed_dim = embed_dim
        vision_heads = vision_width // 64
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_heads, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = _Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = _Affine(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.initialize_parameters()
        self.constrastive_overbatch = constrastive_overbatch
        self.alignment = alignment
        self.multiattention = multiattention      # stored, never read: as in the reference
        from .losses import cross_entropy
        self.loss_func = cross_entropy            # the reference forgets to define it
        # device state (built lazily on the first forward on a GPU)
        self._flat: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._pool = _Pool()
        self._trigger = None
        self._versions = None
        self.grad_sync = None                     # optional callable(model, tower_name) for DP overlap
        # text tower on the live tokens only (rows after a caption's EOT are dead under the causal mask); results
        # are unchanged, see functional.text_packing.  CE_TEXT_PACK=0 keeps the dense [n, 77] layout.
        self.pack_text = os.environ.get("CE_TEXT_PACK", "1") != "0"
        # fp8 (OCP e4m3) operand path for the blocks' Linear GEMMs (BASELINE config 5; which tensors may go to low
        # precision follows convert_weights, model_clip.py:554-575): bit 0 = forward GEMMs, bit 1 = input-gradient
        # GEMMs.  fp32 masters, bf16 copies (weight gradients) and everything else are unchanged.  Off by default.
        self.fp8 = int(os.environ.get("CE_FP8", "0"))
        # Residual stream and gradient stream of both towers in IEEE fp16 instead of fp32 (include/clip_event_hip.h,
        # CE_T_F16): the stream is read and written four times per block and direction and never enters a matrix unit,
        # so this halves 40 % of a block's HBM traffic.  fp16's 11 significand bits keep the gradient noise floor where
        # the bf16 GEMM operands put it (tests/stream16_emulation.py).  The gradient stream is stored multiplied by a
        # power of two chosen per backward pass so that the largest element of the gradient entering the tower sits at
        # ``grad_target`` (ce_grad_scale; fp16 stores saturate at 65504, so 65504 / grad_target = 1024x is the growth the
        # gradient may see on its way down: measured x17-34 in the text tower, x1.1-1.3 in the image tower, tools/diag/grad_growth.py).  CE_STREAM16=0 keeps both streams in fp32 as the reference does.
        self.stream16 = os.environ.get("CE_STREAM16", "1") != "0"
        self.grad_target = float(os.environ.get("CE_GRAD_TARGET", "64"))

    # ---- copy / pickle: the device-side tables (ctypes descriptors, workspace pool, streams, operand copies) are
    # rebuilt lazily by _prepare(); only the parameters and the plain attributes travel --------------------------
    _RUNTIME = ("_flat", "_flat_grad", "_flat16", "_offsets", "_ranges", "_layer_end", "_pmap", "_pool", "_trigger",
                "_versions", "_w16", "_w16t", "_kp", "_kp_real", "_conv_pad", "_conv_gpad", "_cast_list", "_tjobs",
                "_tjobs_bwd", "_adam_tiles_ok", "_wt_fresh", "_wt_event", "_aux_stream", "_mirror_fresh", "_mirror_versions", "_vdesc", "_tdesc", "_vblocks", "_tblocks",
                "_side_streams", "_main_stream", "_pack_cache", "_cls_rows", "_sat", "_sat_poll", "_step_events", "grad_sync", "_w8", "_fp8_fresh", "_zero_table", "_zero_tables", "_adam_segs", "_first_touch")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._RUNTIME:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._flat = None
        self._flat_grad = None
        self._pool = _Pool()
        self._trigger = None
        self._versions = None
        self.grad_sync = None
        for p in self.parameters():            # parameters arrive as copies of the views: gradients start afresh
            p.grad = None

    # ---- reference API -------------------------------------------------------------------
    def set_hyps(self, constrastive_overbatch=True, alignment=False, multiattention=False):
        self.constrastive_overbatch = constrastive_overbatch
        self.alignment = alignment
        self.multiattention = multiattention

    def initialize_parameters(self):
        """model_clip.py:348-375 (text tower only; the vision tower keeps its constructor init)."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        w, layers = self.transformer.width, self.transformer.layers
        proj_std = (w ** -0.5) * ((2 * layers) ** -0.5)
        attn_std = w ** -0.5
        fc_std = (2 * w) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    # ---- flat parameter / gradient buffers ---------------------------------------------------
    def _flat_ok(self) -> bool:
        if self._flat is None:
            return False
        p = self.positional_embedding
        return p.device == self._flat.device and p.data_ptr() == self._flat.data_ptr() + self._offsets["positional_embedding"] * 4

    def _prepare(self):
        """Move every parameter into one flat fp32 buffer (views keep the reference's names)."""
        dev = self.positional_embedding.device
        if dev.type != "cuda":
            raise L.HipExtensionMissing("clip_event_amd computes on an AMD GPU only; move the model with .cuda() first")
        lib()
        named = list(self.named_parameters())
        offsets, off = {}, 0
        pmap = dict(named)
        # Contiguous range per tower, and inside a tower the parameters in the order their gradients become final
        # during the backward (output side first, residual blocks top-down, input embeddings last): the gradients
        # of "everything above block l" are then one contiguous prefix of the tower's range, which lets the
        # data-parallel all-reduce start on it while the lower blocks are still running (distributed.GradSync).
        text_side = ("token_embedding.", "positional_embedding", "ln_final.", "text_projection")

        def block_of(n):
            parts = n.split(".")
            return int(parts[parts.index("resblocks") + 1]) if "resblocks" in parts else None

        def tower_order(names, first, last_):
            head = [n for n in names if block_of(n) is None and any(n.startswith(f) for f in first)]
            tail = [n for n in names if block_of(n) is None and any(n.startswith(f) for f in last_)]
            rest = [n for n in names if block_of(n) is None and n not in head and n not in tail]
            if rest:
                raise RuntimeError(f"unplaced parameters {rest}")
            blocks = sorted({block_of(n) for n in names if block_of(n) is not None}, reverse=True)
            return head, [[n for n in names if block_of(n) == b] for b in blocks], blocks, tail

        groups = {
            "head": ([n for n, _ in named if not n.startswith(("visual.", "transformer.") + text_side)], [], [], []),
            "visual": tower_order([n for n, _ in named if n.startswith("visual.")],
                                  ("visual.ln_post.", "visual.proj"),
                                  ("visual.ln_pre.", "visual.conv1.", "visual.positional_embedding", "visual.class_embedding")),
            "text": tower_order([n for n, _ in named if n.startswith(("transformer.",) + text_side)],
                                ("ln_final.", "text_projection"), ("positional_embedding", "token_embedding.")),
        }
        ranges, layer_end = {}, {}
        # every boundary a gradient piece can end on (tower start / end, the end of every block) is a multiple of 8 x 64 elements:
        # a piece then splits into 2, 4 or 8 equal shards of whole 64-element groups (distributed.ShardPlan: reduce-scatter +
        # sharded Adam + all-gather of the bf16 mirror)
        PIECE = 512
        for group in ("head", "visual", "text"):
            head, per_block, blocks, tail = groups[group]
            off = (off + PIECE - 1) // PIECE * PIECE
            start = off
            layer_end[group] = {}
            for n in head:
                offsets[n] = off
                off += (pmap[n].numel() + 63) // 64 * 64
            for b, names_b in zip(blocks, per_block):
                for n in names_b:
                    offsets[n] = off
                    off += (pmap[n].numel() + 63) // 64 * 64
                off = (off + PIECE - 1) // PIECE * PIECE
                layer_end[group][b] = off          # gradients of blocks >= b (and the output side) end here
            for n in tail:
                offsets[n] = off
                off += (pmap[n].numel() + 63) // 64 * 64
            off = (off + PIECE - 1) // PIECE * PIECE
            ranges[group] = (start, off)
        if len(offsets) != len(named):
            raise RuntimeError("flat layout lost a parameter")
        self._layer_end = layer_end
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p in named:
                if p.dtype != torch.float32:
                    raise RuntimeError(f"parameter {n} is {p.dtype}; master weights are fp32")
                view = flat[offsets[n]: offsets[n] + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
        self._flat, self._flat_grad, self._offsets, self._ranges = flat, flat_grad, offsets, ranges
        self._pmap = pmap
        self._trigger = torch.zeros(1, device=dev, requires_grad=True)
        # fp16-stream saturation counters (include/clip_event_hip.h, ce_stream16_set_counters): [forward stream, gradient stream]
        self._sat = L.sat_counters(dev)
        self._sat_poll = None
        self._build_device_tables()
        self._versions = None

    def _gview(self, name: str) -> torch.Tensor:
        p = self._pmap[name]
        o = self._offsets[name]
        return self._flat_grad[o: o + p.numel()].view(p.shape)

    def _attach_grads(self):
        """Point every ``p.grad`` at its slice of the flat gradient buffer.  A slice whose
        ``p.grad`` was None (``zero_grad(set_to_none=True)``) is zero-filled first; a foreign
        ``p.grad`` tensor (autograd accumulated into a parameter before our backward ran, e.g.
        ``logit_scale``) is copied into its slice."""
        base = self._flat_grad.data_ptr()
        todo = [(n, p) for n, p in self._pmap.items() if p.grad is None or p.grad.data_ptr() != base + self._offsets[n] * 4]
        if not todo:
            return
        if all(p.grad is None for _, p in todo) and len(todo) == len(self._pmap):
            self._flat_grad.zero_()
            for n, p in todo:
                p.grad = self._gview(n)
        else:
            for n, p in todo:
                view = self._gview(n)
                if p.grad is None:
                    view.zero_()
                else:
                    view.copy_(p.grad)
                p.grad = view
        # The fills above ran on the stream of whichever backward node came first (a tower's side stream).  The
        # other tower's backward writes the same buffer from ITS stream: order it behind the fills, or its first
        # weight gradients land before the zero-fill and are wiped.
        cur = torch.cuda.current_stream()
        for s_ in (getattr(self, "_side_streams", None) or ()):
            if s_ != cur:
                s_.wait_stream(cur)
        ms = getattr(self, "_main_stream", None)
        if ms is not None and ms != cur:
            ms.wait_stream(cur)

    # ---- first-touch weight gradients ------------------------------------------------------------------------
    # The four Linear weights of every residual block are 85 % of the gradient buffer, and each receives its gradient
    # from exactly one (grouped) launch per tower pass.  A step that starts with ``zero_grad_first_touch`` zero-fills
    # only the OTHER tensors; the first backward pass of each tower then WRITES its block weight gradients
    # (ce_tower_desc.wgrad_overwrite: plain stores from unsplit tiles instead of float atomics), later passes of the
    # same step accumulate.  Until that first pass has run, ``.grad`` of those weights holds the previous step's
    # values, so only code that owns the whole step (engine.train_step with FusedAdam) takes this path;
    # ``zero_grad()`` keeps torch's contract.
    _BLOCK_WEIGHTS = ("attn.in_proj_weight", "attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")

    def _is_block_weight(self, name: str) -> bool:
        return "resblocks." in name and name.endswith(self._BLOCK_WEIGHTS)

    def _zero_table_for(self):
        """Chunk table of the gradient segments a first-touch step zero-fills: every element outside the block weights."""
        if getattr(self, "_zero_tables", None) is None:
            keep = sorted((self._offsets[n], self._offsets[n] + self._pmap[n].numel()) for n in self._pmap if self._is_block_weight(n))
            segs, pos = [], 0
            for lo, hi in keep:
                if lo > pos:
                    segs.append((pos, lo))
                pos = (hi + 63) // 64 * 64          # the padding behind a tensor is never read
            if pos < self._flat_grad.numel():
                segs.append((pos, self._flat_grad.numel()))
            chunks = []
            for lo, hi in segs:
                for c in range(lo, hi, 1 << 16):
                    chunks.append((c, min(c + (1 << 16), hi)))
            self._zero_tables = torch.tensor(chunks, dtype=torch.int64).reshape(-1, 2).to(self._flat_grad.device)
        return self._zero_tables

    def _adam_segment_table(self):
        """The zero-fill chunk table (= every element outside the block weights) cut into pieces of at most 2048 elements: one
        workgroup of ``ce_adam_step_tiles``' segment kernel each."""
        if getattr(self, "_adam_segs", None) is None:
            rows = []
            for lo, hi in self._zero_table_for().tolist():
                for c in range(lo, hi, 2048):
                    rows.append((c, min(c + 2048, hi)))
            self._adam_segs = torch.tensor(rows, dtype=torch.int64).reshape(-1, 2).to(self._flat_grad.device)
        return self._adam_segs

    def _zero_segments(self, table):
        if table.shape[0]:
            check(lib().ce_zero_segments(ptr(self._flat_grad), ptr(table), c_int(table.shape[0]), stream()), "ce_zero_segments")

    def zero_grad_first_touch(self):
        if self._flat_grad is None or os.environ.get("CE_WGRAD_FIRST_TOUCH", "1") == "0":
            return self.zero_grad()
        self._zero_segments(self._zero_table_for())
        self._attach_grads_fast()
        self._first_touch = {"visual", "text"}

    def _settle_first_touch(self):
        """A tower whose backward did not run since ``zero_grad_first_touch``: its block weight gradients were neither
        zeroed nor written -- zero them now (called before the gradients are consumed)."""
        pending = getattr(self, "_first_touch", None)
        if not pending:
            return
        for n in self._pmap:
            if self._is_block_weight(n) and (("visual" in pending and n.startswith("visual.")) or
                                             ("text" in pending and not n.startswith("visual."))):
                self._gview(n).zero_()
        self._first_touch = set()

    def zero_grad(self, set_to_none: bool = False):  # noqa: D401 - nn.Module API
        self._first_touch = set()
        if self._flat_grad is not None and not set_to_none:
            self._flat_grad.zero_()
            self._attach_grads_fast()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def _attach_grads_fast(self):
        for n, p in self._pmap.items():
            if p.grad is None or p.grad.data_ptr() != self._flat_grad.data_ptr() + self._offsets[n] * 4:
                p.grad = self._gview(n)

    # ---- fp16 residual / gradient stream: saturation telemetry ------------------------------------------------
    def stream16_saturation(self, reset: bool = False):
        """``(forward, gradient)`` counts of fp16-stream values that hit the +-65504 clamp since the counters were last
        reset (sticky device counters fed by every LayerNorm launch; synchronises).  Non-zero = some step computed with a
        clipped activation or gradient -- something the reference's non-finite-loss exit (engine.py:79-82) cannot see."""
        if getattr(self, "_sat", None) is None:
            return (0, 0)
        f, g = (int(v) for v in self._sat.cpu())
        if reset:
            self._sat.zero_()
        return (f, g)

    def poll_stream16_saturation(self):
        """The same check without a host synchronisation: starts an asynchronous copy of the counters and examines the copy
        the PREVIOUS call started (by then long complete).  Raises ``Stream16Saturation`` once a copy shows a clamp."""
        if getattr(self, "_sat", None) is None or not self.stream16:
            return
        prev = self._sat_poll
        if prev is not None and prev[1].query():
            f, g = (int(v) for v in prev[0])
            self._sat_poll = prev = None
            if f or g:
                raise Stream16Saturation(f, g)
        if prev is None:
            host = torch.empty(2, dtype=torch.int32, pin_memory=True)
            host.copy_(self._sat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._sat_poll = (host, ev)

    # ---- bf16 operand copies + tower descriptors -------------------------------------------
    def _build_device_tables(self):
        dev = self._flat.device
        self._w16: Dict[str, torch.Tensor] = {}
        self._w16t: Dict[str, torch.Tensor] = {}
        # bf16 mirror of the whole flat parameter buffer (same offsets): the straight [out,in] operand copies
        # are views of it; the fused Adam kernel writes it together with the fp32 masters.
        self._flat16 = torch.empty(self._flat.numel(), dtype=torch.bfloat16, device=dev)

        def view16(n, shape=None):
            p = self._pmap[n]
            o = self._offsets[n]
            return self._flat16[o: o + p.numel()].view(p.shape if shape is None else shape)

        gemm_names = [n for n in self._pmap if n.endswith(_GEMM_SUFFIXES)]
        for n in gemm_names:
            p = self._pmap[n]
            self._w16[n] = view16(n)
            self._w16t[n] = torch.empty(p.shape[1], p.shape[0], dtype=torch.bfloat16, device=dev)
        v = self.visual
        kp = 3 * v.patch_size * v.patch_size
        self._kp_real = kp
        if kp % 8 == 0:
            self._kp = kp
            self._conv_pad = None
            self._w16["visual.conv1.weight"] = view16("visual.conv1.weight", (self.vision_width, kp))
        else:
            # patch 14 (ViT-L/14): 588 input columns.  The GEMM operands need 16-byte rows, so the patch matrix and
            # a private bf16 copy of the weight are zero-padded to a multiple of 64 columns; the weight gradient is
            # formed in a padded fp32 scratch and its real columns are added into the flat gradient buffer.
            self._kp = (kp + 63) // 64 * 64
            self._conv_pad = torch.zeros(self.vision_width, self._kp, dtype=torch.bfloat16, device=dev)
            self._conv_gpad = torch.zeros(self.vision_width, self._kp, dtype=torch.float32, device=dev)
            self._w16["visual.conv1.weight"] = self._conv_pad
        for n in ("visual.proj", "text_projection"):
            p = self._pmap[n]
            self._w16[n] = view16(n)                                                               # [width, E]
            self._w16t[n] = torch.empty(p.shape[1], p.shape[0], dtype=torch.bfloat16, device=dev)  # [E, width]
        self._cast_list = gemm_names + ["visual.conv1.weight", "visual.proj", "text_projection"]
        # one-launch transposition table for every W^T copy
        from ._lib import TransposeJob
        def table(names):
            jobs = (TransposeJob * len(names))()
            tiles = 0
            for i, n in enumerate(names):
                src, dst = self._w16[n], self._w16t[n]
                jobs[i].src, jobs[i].dst = src.data_ptr(), dst.data_ptr()
                jobs[i].rows, jobs[i].cols, jobs[i].tile_start = src.shape[0], src.shape[1], tiles
                tiles += ((src.shape[0] + 63) // 64) * ((src.shape[1] + 63) // 64)
            return torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev), len(names), tiles

        # two tables: the feature projections' W^T are FORWARD operands; the blocks' W^T are read by the backward only
        # (input-gradient GEMMs), so their rebuild can run beside the next forward (refresh_operands)
        self._tjobs = table(["visual.proj", "text_projection"])
        self._tjobs_bwd = table(gemm_names)
        # fused Adam in tiles (optim.FusedAdam.step -> ce_adam_step_tiles) leaves these W^T copies behind itself: possible when the
        # table is exactly the block weights (whose complement is the zero-fill chunk table) and every matrix has 8-multiples
        self._adam_tiles_ok = (set(gemm_names) == {n for n in self._pmap if self._is_block_weight(n)} and
                               all(self._pmap[n].shape[0] % 8 == 0 and self._pmap[n].shape[1] % 8 == 0 for n in gemm_names))
        self._wt_fresh = False            # True when the fused Adam has just written the blocks' W^T copies as well
        self._wt_event = None
        self._mirror_fresh = False        # True when the Adam kernel has just written _flat16 ...
        self._mirror_versions = None      # ... from masters at these parameter versions

        def desc(prefix: str, tr: Transformer, tokens: int, causal: bool):
            arr = (_BlockParams * tr.layers)()
            for i in range(tr.layers):
                b = f"{prefix}resblocks.{i}."
                f = arr[i]

                def P(name):
                    return self._pmap[b + name].data_ptr()

                def G(name):
                    return self._gview(b + name).data_ptr()

                f.ln1_w, f.ln1_b, f.ln2_w, f.ln2_b = P("ln_1.weight"), P("ln_1.bias"), P("ln_2.weight"), P("ln_2.bias")
                f.b_qkv, f.b_out = P("attn.in_proj_bias"), P("attn.out_proj.bias")
                f.b_fc, f.b_proj = P("mlp.c_fc.bias"), P("mlp.c_proj.bias")
                for short, name in (("qkv", "attn.in_proj_weight"), ("out", "attn.out_proj.weight"),
                                    ("fc", "mlp.c_fc.weight"), ("proj", "mlp.c_proj.weight")):
                    setattr(f, "w_" + short, self._w16[b + name].data_ptr())
                    setattr(f, "wt_" + short, self._w16t[b + name].data_ptr())
                    setattr(f, "g_w_" + short, G(name))
                f.g_ln1_w, f.g_ln1_b, f.g_ln2_w, f.g_ln2_b = G("ln_1.weight"), G("ln_1.bias"), G("ln_2.weight"), G("ln_2.bias")
                f.g_b_qkv, f.g_b_out = G("attn.in_proj_bias"), G("attn.out_proj.bias")
                f.g_b_fc, f.g_b_proj = G("mlp.c_fc.bias"), G("mlp.c_proj.bias")
            d = _TowerDesc(tr.layers, tr.width, tr.heads, tokens, 1 if causal else 0, arr, 0, 0, 0, None)
            d._keep = arr
            return d

        self._vdesc = desc("visual.transformer.", self.visual.transformer, self.visual.patch_num ** 2 + 1, False)
        self._tdesc = desc("transformer.", self.transformer, self.context_length, True)
        lib().ce_tower_workspace_bytes.restype = ctypes.c_size_t
        self._w8 = None
        self._fp8_fresh = False

    def _build_fp8_tables(self):
        """e4m3 copies of the blocks' GEMM weights, one fp32 scale per row: straight [out,in] (scale per output
        channel) for the forward, transposed [in,out] (scale per input channel) for the input-gradient GEMMs."""
        dev = self._flat.device
        self._w8 = {}
        for prefix, d in (("visual.transformer.", self._vdesc), ("transformer.", self._tdesc)):
            for i in range(d.layers):
                f = d.blocks[i]
                for short, name in (("qkv", "attn.in_proj_weight"), ("out", "attn.out_proj.weight"),
                                    ("fc", "mlp.c_fc.weight"), ("proj", "mlp.c_proj.weight")):
                    n = f"{prefix}resblocks.{i}.{name}"
                    o, k = self._pmap[n].shape
                    t = (torch.empty(o, k, dtype=torch.uint8, device=dev), torch.empty(o, dtype=torch.float32, device=dev),
                         torch.empty(k, o, dtype=torch.uint8, device=dev), torch.empty(k, dtype=torch.float32, device=dev))
                    self._w8[n] = t
                    setattr(f, "w8_" + short, t[0].data_ptr())
                    setattr(f, "s8_" + short, t[1].data_ptr())
                    setattr(f, "wt8_" + short, t[2].data_ptr())
                    setattr(f, "st8_" + short, t[3].data_ptr())
        self._fp8_fresh = False
        self._qjobs_key = None

    def _refresh_fp8(self):
        """Requantise from the bf16 operand copies (they have just been refreshed from the masters)."""
        if self._w8 is None:
            self._build_fp8_tables()
        self.wait_transposes()            # with fp8 & 2 the quantiser reads the W^T copies an asynchronous rebuild may still be writing
        cl, s = lib(), stream()
        key = int(self.fp8) & 2
        if getattr(self, "_qjobs_key", None) != key:
            # one launch for every weight (and, with the input-gradient GEMMs in fp8, every transposed copy): a job table in
            # device memory, as for the transposes
            from ._lib import QuantJob
            jobs, groups = [], 0
            for n, (w8, s8, w8t, s8t) in self._w8.items():
                o, k = w8.shape
                todo = [(self._w16[n], w8, s8, o, k)] + ([(self._w16t[n], w8t, s8t, k, o)] if key else [])
                for src, dst, sc, rows, cols in todo:
                    if cols > 4096:
                        raise RuntimeError("fp8 weight rows longer than 4096 are not supported")
                    jobs.append(QuantJob(src.data_ptr(), dst.data_ptr(), sc.data_ptr(), cols, cols, rows, cols, groups, 0))
                    groups += (rows + 3) // 4
            arr = (QuantJob * len(jobs))(*jobs)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._qjobs = host.to(self._flat.device)
            self._qjobs_n, self._qjobs_groups, self._qjobs_key = len(jobs), groups, key
        check(cl.ce_quant_rows_fp8_multi(ptr(self._qjobs), c_int(self._qjobs_n), c_int(self._qjobs_groups), s),
              "ce_quant_rows_fp8_multi")
        self._fp8_fresh = True

    def refresh_operands(self, force: bool = False):
        """Re-cast the bf16 GEMM operands from the fp32 masters when a master changed
        (in-place optimiser update, ``load_state_dict``)."""
        vers = tuple(self._pmap[n]._version for n in self._cast_list)
        self._vdesc.fp8 = self._tdesc.fp8 = int(self.fp8)
        self._vdesc.stream16 = self._tdesc.stream16 = 1 if self.stream16 else 0
        if not force and vers == self._versions:
            if self.fp8 and not self._fp8_fresh:
                self._refresh_fp8()
            return
        s = stream()
        cl = lib()
        # the bf16 mirror written by the fused Adam kernel is only as good as the masters it was cast from: if any
        # master has moved since (load_state_dict, a stock optimiser step, an EMA swap), cast again
        if not (self._mirror_fresh and vers == self._mirror_versions):
            self.wait_transposes()        # an earlier asynchronous rebuild may still be reading the mirror
            # masters changed outside the fused optimiser: rebuild the whole bf16 mirror (one launch)
            check(cl.ce_cast_bf16(ptr(self._flat), ptr(self._flat16), c_long(self._flat.numel()), s), "ce_cast_bf16")
        tj, tn_, tt = self._tjobs
        check(cl.ce_multi_transpose_bf16(ptr(tj), c_int(tn_), c_int(tt), s), "ce_multi_transpose_bf16")
        tj, tn_, tt = self._tjobs_bwd
        if self._mirror_fresh and vers == self._mirror_versions and getattr(self, "_wt_fresh", False):
            pass                          # the fused Adam wrote the blocks' W^T copies with the update (ce_adam_step_tiles)
        elif getattr(self, "tower_streams", True) and not self.fp8 and os.environ.get("CE_ASYNC_TRANSPOSE", "1") != "0":
            # the blocks' W^T copies (0.15 ms of pure copying) on a third stream, beside the forward that is about to be
            # enqueued; every tower backward, and whoever rewrites the bf16 mirror next, waits for the event
            cur = torch.cuda.current_stream()
            if getattr(self, "_aux_stream", None) is None or self._aux_stream.device != self._flat.device:
                self._aux_stream = torch.cuda.Stream(device=self._flat.device)
            self._aux_stream.wait_stream(cur)
            with torch.cuda.stream(self._aux_stream):
                check(cl.ce_multi_transpose_bf16(ptr(tj), c_int(tn_), c_int(tt), stream()), "ce_multi_transpose_bf16(blocks)")
                self._wt_event = torch.cuda.Event()
                self._wt_event.record(self._aux_stream)
        else:
            self.wait_transposes()
            check(cl.ce_multi_transpose_bf16(ptr(tj), c_int(tn_), c_int(tt), s), "ce_multi_transpose_bf16(blocks)")
        if self._conv_pad is not None:
            check(cl.ce_cast_transpose(ptr(self._pmap["visual.conv1.weight"]), ptr(self._conv_pad), c_long(self._kp), None,
                                       c_long(0), c_int(self.vision_width), c_int(self._kp_real), s), "ce_cast_transpose(conv1)")
        self._mirror_fresh = False
        self._wt_fresh = False
        self._versions = vers
        self._fp8_fresh = False
        if self.fp8:
            self._refresh_fp8()

    def wait_transposes(self):
        """Order the current stream behind the asynchronous rebuild of the blocks' W^T copies (refresh_operands)."""
        ev = getattr(self, "_wt_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def mark_operands_stale(self, mirror_fresh: bool = False, wt_fresh: bool = False):
        """``mirror_fresh``: the caller (fused Adam) has already written the bf16 mirror of the new masters,
        only the transposed copies need rebuilding; ``wt_fresh``: it has written the blocks' W^T copies too."""
        self._versions = None
        self._mirror_fresh = mirror_fresh
        self._wt_fresh = bool(mirror_fresh and wt_fresh)
        self._mirror_versions = tuple(self._pmap[n]._version for n in self._cast_list) if mirror_fresh else None

    def _ready(self):
        if not self._flat_ok():
            self._prepare()
        dev = self._flat.device
        if torch.cuda.current_device() != dev.index:
            # launches go to the CURRENT device's stream: running a model that lives on another GPU would fault
            raise RuntimeError(f"the model lives on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                               "call torch.cuda.set_device() first (one process per GPU)")
        self.refresh_operands()

    # ---- encoders -------------------------------------------------------------------------
    def _note_pass(self, tower: str):
        """Bookkeeping before a tower forward: refuse torch's DistributedDataParallel (its reducer waits for
        per-parameter autograd hooks that the HIP backward never fires -- an "unchanged" train.py would die on the
        second iteration inside the reducer with an unrelated message) and tell the gradient exchange that one more
        backward will write this tower's gradient range."""
        from torch.nn.parallel import DistributedDataParallel as TorchDDP
        if getattr(TorchDDP, "_active_ddp_module", None) is not None:
            raise RuntimeError(
                "clip_event_amd.CLIP cannot run inside torch.nn.parallel.DistributedDataParallel: parameter gradients "
                "are written by the HIP backward as side effects, so DDP's autograd hooks never fire.  Wrap the model "
                "with clip_event_amd.distributed.DistributedDataParallel(model, device_ids=[gpu]) instead (same call "
                "site, train.py:222-225; see INTEGRATION.md).")
        if self.grad_sync is not None and torch.is_grad_enabled() and hasattr(self.grad_sync, "note_forward"):
            self.grad_sync.note_forward(tower)

    def encode_image(self, image, use_grid: bool = False):
        """model_clip.py:390-391 -> VisualTransformer.forward (:232-263)."""
        self._ready()
        self._note_pass("visual")
        from .functional import EncodeImageFn
        return EncodeImageFn.apply(image, self._trigger, self, bool(use_grid))

    def encode_text(self, text, lengths=None):
        """model_clip.py:398-417.  ``lengths`` (optional, host integers: tokens up to and including each caption's EOT) spares
        the text tower its one device read-back per new token tensor; ``functional.attach_lengths`` tags a tensor instead."""
        self._ready()
        self._note_pass("text")
        from .functional import EncodeTextFn, attach_lengths
        if lengths is not None:
            attach_lengths(text, lengths)
        return EncodeTextFn.apply(text, self._trigger, self)

    def encode_both(self, image, text, use_grid: bool = False):
        """Image and text towers of one step.  They are independent until the logits, so with
        ``tower_streams`` (default) they run on two HIP streams: the ramp-up / tail of every kernel of one
        tower (and the CUs a ragged tile grid leaves idle) is filled by the other tower's kernels.  The
        autograd engine replays each tower's backward on its forward stream, so the backward overlaps too."""
        self._ready()
        if not getattr(self, "tower_streams", True):
            self._main_stream = None
            return self.encode_image(image, use_grid), self.encode_text(text)
        if getattr(self, "_side_streams", None) is None or self._side_streams[0].device != self._flat.device:
            # the image tower is the longer chain: its stream gets the higher priority, the text tower's kernels fill
            # in around it (measured 0.7 % on the step; CE_IMG_STREAM_PRIORITY / CE_TXT_STREAM_PRIORITY override)
            pri = int(os.environ.get("CE_IMG_STREAM_PRIORITY", "-1"))
            self._side_streams = (torch.cuda.Stream(device=self._flat.device, priority=pri),
                                  torch.cuda.Stream(device=self._flat.device, priority=int(os.environ.get("CE_TXT_STREAM_PRIORITY", "0"))))
        s_img, s_txt = self._side_streams
        cur = torch.cuda.current_stream()
        self._main_stream = cur
        s_img.wait_stream(cur)
        s_txt.wait_stream(cur)
        with torch.cuda.stream(s_img):
            image_features = self.encode_image(image, use_grid)
        with torch.cuda.stream(s_txt):
            text_features = self.encode_text(text)
        cur.wait_stream(s_img)
        cur.wait_stream(s_txt)
        image_features.record_stream(cur)
        text_features.record_stream(cur)
        return image_features, text_features

    def forward(self, image, text, train_arg=None, bboxs=None, bbox_desc_vec=None, bbox_label_vec=None):
        """model_clip.py:419-528."""
        from .functional import logits_from_features
        if train_arg is None:
            image_features, text_features = self.encode_both(image, text)
            return logits_from_features(image_features, text_features, self.logit_scale, self.constrastive_overbatch)
        image_features, text_features, loss_per_bbox, loss_per_arg = self.encode_with_regions(
            image, text, train_arg, bboxs, bbox_desc_vec, bbox_label_vec)
        logits_per_image, logits_per_text = logits_from_features(
            image_features, text_features, self.logit_scale, self.constrastive_overbatch)
        return logits_per_image, logits_per_text, loss_per_bbox, loss_per_arg

    def encode_with_regions(self, image, text, train_arg, bboxs, bbox_desc_vec, bbox_label_vec):
        """The ``train_arg`` half of model_clip.py:419-488: both towers (image tower with its patch grid), then the
        region / argument losses from the grid.  Returns ``(image_features [B,E], text_features, loss_per_bbox,
        loss_per_arg)``; the caller forms the logits (locally, or over the all-gathered batch)."""
        from .region import region_losses
        image_features, text_features = self.encode_both(image, text, use_grid=True)
        B = image_features.size(0)
        pn = self.visual.patch_num
        grid_features = image_features[:, 1:, :].reshape(B, pn, pn, -1)
        image_features = image_features[:, 0, :]
        loss_per_bbox, loss_per_arg = region_losses(self, grid_features, bboxs, bbox_desc_vec, bbox_label_vec, train_arg)
        return image_features, text_features, loss_per_bbox, loss_per_arg

    def sim_entity(self, img_obj, txt_ent):
        """model_clip.py:531-552: un-normalised object / entity features."""
        B, n_img, n_txt = img_obj.size(0), img_obj.size(1), txt_ent.size(1)
        image_features = self.encode_image(img_obj.reshape(B * n_img, img_obj.size(2), img_obj.size(3), img_obj.size(4)))
        image_features = image_features.view(B, n_img, -1)
        text_features = self.encode_text(txt_ent.reshape(B * n_txt, txt_ent.size(2)), getattr(txt_ent, "_ce_lengths", None))
        text_features = text_features.view(B, n_txt, -1)
        return image_features, text_features


def build_model(state_dict: dict) -> CLIP:
    """model_clip.py:578-617 (ViT branch): hyper-parameters inferred from tensor shapes, strict load,
    returned in train mode."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("ModifiedResNet checkpoints are out of scope (ViT towers only)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict.keys() if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers)
    for key in ["input_resolution", "context_length", "vocab_size"]:
        if key in state_dict:
            del state_dict[key]
    model.load_state_dict(state_dict)
    return model
