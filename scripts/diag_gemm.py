import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
fn = _C.lib.hypad_diag_gemm
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
W = torch.randn(1024 * 256, device="cuda")
out = torch.zeros(4, dtype=torch.int64, device="cuda")
print("kind K N mt threads : cycles(first) cycles(4th)  | tiles/wave  MFMA-cycles/wave")
for kind, K, N in ((0, 128, 192), (0, 128, 384), (0, 128, 128), (0, 128, 64), (0, 64, 192), (0, 32, 192), (0, 100, 100), (0, 20, 20), (0, 50, 192),
                   (1, 128, 384), (1, 128, 192), (1, 100, 100), (1, 20, 20)):
    for mt in (1, 2):
        for threads in (256, 512, 1024):
            rc = fn(_C.ptr(W), K, N, mt, kind, threads, _C.ptr(out), _C.stream()); assert rc == 0
            torch.cuda.synchronize()
            nw = threads // 64
            if kind == 0:
                tiles = -(-((N + 15) // 16) // nw); mf = tiles * (-(-K // 4)) * mt * 32
            else:
                tiles = 0; mf = 0
            print(f"{'nt' if kind == 0 else 'nn'} K={K:4d} N={N:4d} mt={mt} thr={threads:5d}: {out[0].item():7d} {out[1].item():7d} | {tiles} {mf}")
