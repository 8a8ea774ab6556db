"""Do fp32 MFMAs of one wave and VALU / transcendental / LDS work of another wave on the same SIMD overlap?  (development library)"""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"
import ctypes
import torch
from hypad_amd import _C
fn = _C.lib.hypad_diag_coissue
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(16, dtype=torch.int64, device="cuda")
NM, NV = 64 * 16, 128 * 8
for w2 in range(1, 8):
    out.zero_(); fn(3 | (2 << 2), w2, _C.ptr(out), _C.stream()); torch.cuda.synchronize()
    print(f"MFMA on waves 0 and {w2}: {out[0].item() / NM:.1f} / {out[1].item() / NM:.1f} cycles per MFMA")
W2 = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for kind, name in ((0, "v_fma_f32"), (1, "v_exp_f32"), (2, "MFMA"), (3, "ds_read_b32")):
    res = {}
    for mode in (1, 2, 3, 19):
        for _ in range(3):
            out.zero_()
            fn(mode | (kind << 2), W2, _C.ptr(out), _C.stream())
        torch.cuda.synchronize()
        res[mode] = (out[0].item(), out[1].item())
    nv = NM if kind == 2 else NV
    print(f"wave 4 runs {name:12s}: MFMA wave alone {res[1][0] / NM:6.1f} cyc/MFMA | other alone {res[2][1] / nv:6.1f} cyc/instr | "
          f"together: MFMA {res[3][0] / NM:6.1f} cyc/MFMA, other {res[3][1] / nv:6.1f} cyc/instr | "
          f"other at s_setprio 3: MFMA {res[19][0] / NM:6.1f}, other {res[19][1] / nv:6.1f}")

for kind, name in ((0, "v_fma_f32"), (1, "v_exp_f32")):
    row = []
    for nv in (0, 1, 2, 4, 6, 8):
        for _ in range(3):
            out.zero_(); fn(1 | 32 | (kind << 2), nv, _C.ptr(out), _C.stream())
        torch.cuda.synchronize()
        row.append(f"{nv}: {out[0].item() / NM:.1f}")
    print(f"one wave, n x {name} behind every MFMA -> cycles per MFMA  " + "  ".join(row))
