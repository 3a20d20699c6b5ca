"""ctypes binding of libclip_event_hip.so (the C ABI in include/clip_event_hip.h).

The product path has no CPU fallback: if the library is missing, or a call fails, this
raises.  Tensors are passed as raw device pointers (`tensor.data_ptr()`), the stream as
`torch.cuda.current_stream().cuda_stream`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclip_event_hip.so")
_lib = None

EPI_BF16, EPI_F32, EPI_BIAS_BF16, EPI_BIAS_F32, EPI_BIAS_RESID_F32, EPI_BIAS_GELU, EPI_GELUGRAD_BF16, EPI_BIAS_RESID_F16 = range(8)
T_F32, T_BF16, T_F16 = 0, 1, 2          # element types of stream operands (CE_T_* of include/clip_event_hip.h)


class HipExtensionMissing(RuntimeError):
    pass


class QuantJob(ctypes.Structure):
    """``ce_quant_job`` of include/clip_event_hip.h."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("scale", c_void_p), ("lds_", ctypes.c_long), ("ldd", ctypes.c_long),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("group_start", ctypes.c_int), ("pad_", ctypes.c_int)]


class TransposeJob(ctypes.Structure):
    """``ce_transpose_job`` of include/clip_event_hip.h."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("tile_start", ctypes.c_int), ("pad_", ctypes.c_int)]


def lib() -> ctypes.CDLL:
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not found: build it with `python -m clip_event_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ce_last_error.restype = c_char_p
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().ce_last_error().decode()}")


def ptr(t) -> c_void_p:
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_SAT = {}


def sat_counters(device) -> torch.Tensor:
    """The process-wide device buffer of the fp16-stream saturation counters (``ce_stream16_set_counters``): int32
    [forward stream, gradient stream].  One process drives one GPU, so one buffer, registered once and never freed (the
    library keeps the raw pointer)."""
    key = str(device)
    t = _SAT.get(key)
    if t is None:
        t = _SAT[key] = torch.zeros(2, dtype=torch.int32, device=device)
        check(lib().ce_stream16_set_counters(ptr(t)), "ce_stream16_set_counters")
    return t
