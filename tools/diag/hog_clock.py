"""Shader clock under (a) nothing, (b) the training step, as seen by a 1-wave hog; and the step time with / without it."""
import ctypes, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clip_event_amd._lib import lib
cl = lib()
cl.ce_cu_hog_clock_mhz.restype = ctypes.c_double
s = torch.cuda.current_stream()
cl.ce_cu_hog(ctypes.c_int(1), ctypes.c_float(20000.0), ctypes.c_void_p(s.cuda_stream))
print("idle chip, 20 ms hog alone:", cl.ce_cu_hog_clock_mhz(), "MHz")
