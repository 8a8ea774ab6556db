import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
fn = _C.lib.hypad_diag_load
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
W = torch.randn(8 * 4 * 16 * 128 * 4, device="cuda")
out = torch.zeros(2 * 16 * 32, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
for mode in (0, 1):
    for threads in (64, 512):
        fn(_C.ptr(W), 128, 4, mode, threads, _C.ptr(out), _C.ptr(sink), _C.stream()); torch.cuda.synchronize()
        o = out.view(2, 16, 32)[:, : threads // 64, :4].cpu()
        print(f"mode={mode} threads={threads} cold pass wave0 per-tile cycles {o[0,0].tolist()}  warm pass {o[1,0].tolist()}  warm mean all waves {o[1].float().mean():.0f}")
