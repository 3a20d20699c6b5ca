"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the reference's image preprocessing, clip.py:62-69:

    Compose([Resize(n_px, interpolation=Image.BICUBIC), CenterCrop(n_px), convert("RGB"), ToTensor(),
             Normalize((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))])

and of the object-patch path dataset_voa.py:222-233 (``image.crop(bbox)`` followed by the same transform).

``torchvision`` is not in this image, so its three transforms are restated from their documented behaviour
(torchvision 0.x ``Resize(int)``: shorter side -> n_px, longer side ``int(n_px * long / short)``, no-op when the
shorter side already equals n_px; ``CenterCrop``: offsets ``int(round((size - n_px) / 2.0))`` with Python's
round-half-even; ``ToTensor``: ``/255`` in fp32; ``Normalize``: ``(x - mean) / std`` in fp32).  The resampling
itself is Pillow's (``Image.resize(..., BICUBIC)``, Pillow 12.2 ``src/libImaging/Resample.c``): separable, the
HORIZONTAL pass first, 8-bit intermediate, coefficients in double -> normalised -> 22-bit fixed point.  Pillow IS
importable here, so ``tests/test_preprocess.py`` pins every function below against PIL itself bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

MEAN = np.array((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
STD = np.array((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)
PRECISION_BITS = 32 - 8 - 2


def bicubic(x: float) -> float:
    """Resample.c ``bicubic_filter`` (a = -0.5)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the whole-image box: per output index the
    first source index, the tap count and the fixed-point taps."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img: np.ndarray, ow: int, oh: int) -> np.ndarray:
    """``Image.resize((ow, oh), BICUBIC)`` on an HWC uint8 array: horizontal pass, then vertical, 8-bit between."""
    h, w, _ = img.shape
    x = img
    if ow != w:
        bounds, kk = precompute_coeffs(w, ow)
        out = np.empty((h, ow, 3), dtype=np.uint8)
        src = x.astype(np.int64)
        for xx in range(ow):
            x0, n = bounds[xx]
            acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            out[:, xx, :] = _clip8(acc)
        x = out
    if oh != h:
        bounds, kk = precompute_coeffs(h, oh)
        out = np.empty((oh, x.shape[1], 3), dtype=np.uint8)
        src = x.astype(np.int64)
        for yy in range(oh):
            y0, n = bounds[yy]
            acc = (1 << (PRECISION_BITS - 1)) + (src[y0:y0 + n, :, :] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0)
            out[yy] = _clip8(acc)
        x = out
    return x


def resized_size(w: int, h: int, n_px: int):
    """torchvision ``Resize(n_px)`` output size ``(ow, oh)``."""
    if (w <= h and w == n_px) or (h <= w and h == n_px):
        return w, h
    if w < h:
        return n_px, int(n_px * h / w)
    return int(n_px * w / h), n_px


def crop_offsets(ow: int, oh: int, n_px: int):
    """torchvision ``CenterCrop``: ``(left, top)``; Python ``round`` (half to even)."""
    return int(round((ow - n_px) / 2.0)), int(round((oh - n_px) / 2.0))


def transform(img: np.ndarray, n_px: int = 224, roi=None) -> np.ndarray:
    """clip.py:62-69 on an HWC uint8 RGB array (optionally on ``img.crop(roi)``, roi = (x0, y0, x1, y1) inside the
    image): float32 [3, n_px, n_px]."""
    if roi is not None:
        x0, y0, x1, y1 = roi
        img = img[y0:y1, x0:x1]
    h, w, _ = img.shape
    ow, oh = resized_size(w, h, n_px)
    if ow < n_px or oh < n_px:
        raise ValueError("CenterCrop padding path (resized side < n_px) cannot occur after Resize(n_px)")
    r = resize_bicubic_u8(img, ow, oh)
    left, top = crop_offsets(ow, oh, n_px)
    c = r[top:top + n_px, left:left + n_px]
    t = c.astype(np.float32) / np.float32(255.0)
    t = (t - MEAN[None, None, :]) / STD[None, None, :]
    return np.ascontiguousarray(t.transpose(2, 0, 1)).astype(np.float32)
