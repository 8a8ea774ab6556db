import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
"""Reproducer sweep for the 16-byte buffer-store hazard (DESIGN.md §4, tile_gemm.h GBuf::st4): stores whose data registers are
rewritten NOPS wait states later, three addressing forms, counted mismatches per 10^6 stores.  Prints one table."""
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
fn = _C.lib.hypad_diag_store16
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
blocks, iters, reps = 1024, 64, 8
buf = torch.empty(blocks * 256 * iters * 4, dtype=torch.int32, device="cuda")
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
names = {0: "s_off offen (scalar row offset)", 1: "0 offen (offset folded into the VGPR)", 2: "s_mov s_off; store s_off offen"}
print(f"{blocks * 256 * iters * reps:,} 16-byte stores per cell; cell = stores that did not land intact")
for form in (0, 1, 2):
    row = []
    for nops in (0, 1, 2, 3, 4):
        bad.zero_()
        for rep in range(reps):
            buf.zero_()
            rc = fn(form, nops, blocks, iters, buf.data_ptr(), bad.data_ptr(), _C.stream())
            assert rc == 0, rc
        torch.cuda.synchronize()
        row.append(int(bad.item()))
    print(f"{names[form]:44s} wait states 0/1/2/3/5 before the overwrite: {row}")
