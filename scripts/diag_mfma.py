import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
fn = _C.lib.hypad_diag_mfma
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(2, dtype=torch.int64, device="cuda")
for variant, nacc, name in ((1, 1, "16x16x4 1 acc"), (2, 2, "16x16x4 2 acc"), (4, 4, "16x16x4 4 acc"), (32, 2, "32x32x2 2 acc")):
    for threads in (64, 256, 512):
        for _ in range(3):
            fn(variant, threads, _C.ptr(out), _C.stream())
        torch.cuda.synchronize()
        n = 256 * 4 * nacc
        print(f"{name:16s} threads={threads}: {out[0].item() / n:.1f} cycles per MFMA per wave")
