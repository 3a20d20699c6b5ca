"""On-device image preprocessing: ``clip._transform`` (clip.py:62-69) and the object-patch path
(dataset_voa.py:195-233) for a batch of decoded uint8 images resident in HBM.

The reference runs PIL + torchvision per image on the host (``num_workers=0``, train.py:212); here the host only
fills one small geometry record per output (sizes are integers known from the image headers) and three HIP
launches do the rest, bit for bit (Pillow's bicubic resampling incl. its 8-bit intermediate and fixed-point taps,
torchvision's size / crop rules, ``/255`` and ``(x - mean) / std`` in correctly rounded fp32).
"""
from __future__ import annotations

import ctypes
import math
from ctypes import c_float, c_int, c_long, c_void_p
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import check, lib, ptr, stream

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


class _Desc(ctypes.Structure):
    """``ce_preproc_desc`` of include/clip_event_hip.h."""
    _fields_ = [("src", c_void_p), ("pitch", c_long), ("x0", c_int), ("y0", c_int), ("w", c_int), ("h", c_int),
                ("ow", c_int), ("oh", c_int), ("left", c_int), ("top", c_int), ("row0", c_int), ("rows", c_int),
                ("tmp_off", c_long)]


def resized_size(w: int, h: int, n_px: int) -> Tuple[int, int]:
    """torchvision ``Resize(n_px)``: shorter side -> n_px, longer ``int(n_px * long / short)``; no-op if already so."""
    if (w <= h and w == n_px) or (h <= w and h == n_px):
        return w, h
    if w < h:
        return n_px, int(n_px * h / w)
    return int(n_px * w / h), n_px


def crop_offsets(ow: int, oh: int, n_px: int) -> Tuple[int, int]:
    """torchvision ``CenterCrop``: ``int(round((size - n_px) / 2.0))`` (round half to even)."""
    return int(round((ow - n_px) / 2.0)), int(round((oh - n_px) / 2.0))


def _bounds(in_size: int, out_size: int, xx: int) -> Tuple[int, int]:
    """First source index and tap count of output index ``xx`` (Pillow ``precompute_coeffs``)."""
    if in_size == out_size:
        return xx, 1
    scale = in_size / out_size
    support = 2.0 * max(scale, 1.0)
    center = (xx + 0.5) * scale
    lo = max(int(center - support + 0.5), 0)
    hi = min(int(center + support + 0.5), in_size)
    return lo, hi - lo


def preprocess(images: Sequence[torch.Tensor], rois: Optional[Sequence[Optional[Sequence[Tuple[int, int, int, int]]]]] = None,
               n_px: int = 224) -> torch.Tensor:
    """``images``: uint8 HWC RGB tensors on the GPU (any sizes).  ``rois`` (optional, per image): list of
    ``(x0, y0, x1, y1)`` boxes inside the image; each image then yields the whole-image tensor FOLLOWED by one tensor
    per box, in that order (dataset_voa.py:195-233).  Returns fp32 ``[n_out, 3, n_px, n_px]``."""
    if not images:
        raise ValueError("preprocess: empty batch")
    dev = images[0].device
    if dev.type != "cuda":
        raise RuntimeError("preprocess runs on the GPU: move the decoded images there first (no CPU fallback)")
    regions = []
    for i, im in enumerate(images):
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or im.device != dev:
            raise ValueError("preprocess: images must be uint8 [H, W, 3] tensors on one GPU")
        if im.stride(2) != 1 or im.stride(1) != 3:
            im = im.contiguous()
            images = list(images)
            images[i] = im
        H, W = int(im.shape[0]), int(im.shape[1])
        boxes = [(0, 0, W, H)] + [tuple(int(v) for v in b) for b in ((rois[i] or []) if rois is not None else [])]
        for (x0, y0, x1, y1) in boxes:
            if not (0 <= x0 < x1 <= W and 0 <= y0 < y1 <= H):
                raise ValueError(f"preprocess: box {(x0, y0, x1, y1)} is not inside the {W}x{H} image")
            regions.append((im, x0, y0, x1 - x0, y1 - y0))
    n_out = len(regions)
    descs = (_Desc * n_out)()
    tmp_bytes, max_rows, max_scale = 0, 1, 1.0
    for o, (im, x0, y0, w, h) in enumerate(regions):
        ow, oh = resized_size(w, h, n_px)
        left, top = crop_offsets(ow, oh, n_px)
        r_first, _ = _bounds(h, oh, top)
        r_last, cnt = _bounds(h, oh, top + n_px - 1)
        rows = r_last + cnt - r_first
        d = descs[o]
        d.src, d.pitch = im.data_ptr(), im.stride(0)
        d.x0, d.y0, d.w, d.h, d.ow, d.oh, d.left, d.top = x0, y0, w, h, ow, oh, left, top
        d.row0, d.rows, d.tmp_off = r_first, rows, tmp_bytes
        tmp_bytes += rows * n_px * 3
        max_rows = max(max_rows, rows)
        max_scale = max(max_scale, w / ow, h / oh)
    kmax = int(math.ceil(2.0 * max_scale)) * 2 + 1
    cl = lib()
    cl.ce_preprocess_table_bytes.restype = ctypes.c_size_t
    table = torch.empty(cl.ce_preprocess_table_bytes(c_int(n_out), c_int(n_px), c_int(kmax)), dtype=torch.uint8, device=dev)
    tmp = torch.empty(max(tmp_bytes, 1), dtype=torch.uint8, device=dev)
    out = torch.empty(n_out, 3, n_px, n_px, dtype=torch.float32, device=dev)
    descs_d = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
    mean = (c_float * 3)(*MEAN)
    std = (c_float * 3)(*STD)
    check(cl.ce_preprocess(ptr(descs_d), c_int(n_out), c_int(n_px), c_int(kmax), c_int(max_rows), ptr(table), ptr(tmp),
                           ptr(out), mean, std, stream()), "ce_preprocess")
    out._keepalive = (descs_d, table, tmp, list(images))   # until the stream has consumed them
    return out
