import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
fn = _C.lib.hypad_diag_tile
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
W = torch.randn(1024 * 256, device="cuda")
out = torch.zeros(64, dtype=torch.int64, device="cuda")
names = ["MFMA only (32)", "direct frag loads + LDS A + MFMA", "direct frag loads only", "LDS A + MFMA", "contiguous -> LDS slab -> MFMA"]
for mode in range(5):
    for threads in (64, 256, 512, 1024):
        fn(_C.ptr(W), mode, threads, _C.ptr(out), _C.stream()); torch.cuda.synchronize()
        o = out[: threads // 64].cpu()
        print(f"{names[mode]:36s} threads={threads:5d}: wave0 {o[0].item():6d}  max {o.max().item():6d}")
