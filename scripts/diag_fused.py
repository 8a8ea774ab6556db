"""Per-stage shader-clock timeline of critic_iteration_kernel (workgroup 0 of each critic), configs[1] shape."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from hypad_amd import _C

dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(128, dtype=torch.int64, device=dev)
fn = _C.lib.hypad_diag_set_fused_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
perm = torch.stack([torch.randperm(bench.N_WINDOWS, device=dev)[: 3 * bench.B] for _ in range(3)]).to(torch.int32).contiguous()
for _ in range(3):
    eng.train_epoch(x, perm, 3, 2, True)
torch.cuda.synchronize()
ms = None
s = st.cpu().numpy().reshape(2, 64)
print("events ms", ms)
for z, nm in ((0, "critic_x"), (1, "critic_z")):
    nh = 4 if z == 0 else 2
    names = ["prologue (record loads, reduce+Adam)", "record->LDS", "fwd + bwd chains in registers (waves 0-2) | Gram (waves 3,4)",
             "ep chain (wave 2) | g + dWrf", "dW gp (+rf of wave 2)", "publish"]
    t = s[z]
    n = len(names)
    d = np.diff(t[: n + 1])
    print(nm, "total cycles", t[40] - t[0], "| prologue: to adam-coef", t[50] - t[0], "barrier", t[51] - t[50], "tiles", t[1] - t[51])
    print("   " + ", ".join(f"{a} {b}" for a, b in zip(names, d)))

    c = s[z]
    marks = [2, 20, 21] + [21 + li for li in range(1, nh)] + [26] + [27 + li for li in range(nh - 2, -1, -1)]
    lab = ["B1", "fwd0 mfma", "fwd0 epi"] + [f"fwd{li}" for li in range(1, nh)] + ["top"] + [f"bwd{li}" for li in range(nh - 2, -1, -1)]
    print("   chain (wave 0): " + ", ".join(f"{lab[i]} {c[marks[i]] - c[marks[i-1]]}" for i in range(1, len(marks))))
