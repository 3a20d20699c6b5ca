"""How long does ce_cu_hog(blocks, us) really last (s_memrealtime rate), alone on the GPU?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clip_event_amd._lib import lib
cl = lib()
s = torch.cuda.current_stream()
for us in (1000.0, 5000.0, 12000.0):
    for k in (1, 8, 32):
        cl.ce_cu_hog(ctypes.c_int(k), ctypes.c_float(us), ctypes.c_void_p(s.cuda_stream))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cl.ce_cu_hog(ctypes.c_int(k), ctypes.c_float(us), ctypes.c_void_p(s.cuda_stream))
        e1.record()
        torch.cuda.synchronize()
        print(f"requested {us:.0f} us on {k} CUs: {e0.elapsed_time(e1) * 1e3:.0f} us")
