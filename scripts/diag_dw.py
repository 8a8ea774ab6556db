"""Per-item timing of dw_adam_kernel (generator table): kind, descriptor, start, duration in shader cycles."""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from hypad_amd import _C

dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(64 + 4 * 2000, dtype=torch.int64, device=dev)
fn = _C.lib.hypad_diag_set_gen_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
for _ in range(5):
    eng.decoder_iteration(x, idx, None, True)
torch.cuda.synchronize()
t = st.cpu().numpy()[64:].reshape(-1, 4)
t = t[t[:, 3] > 0]
t0 = 0
kinds = {0: "weight", 1: "bias", 2: "decay", 3: "ball"}
print("items", len(t), "span cycles", (t[:, 2] + t[:, 3]).max() - t0)
for k in range(4):
    m = t[t[:, 0] == k]
    if len(m):
        print(f"{kinds[k]:6s} n={len(m):4d} dur mean {m[:,3].mean():8.0f} max {m[:,3].max():8d}  start mean {(m[:,2]-t0).mean():8.0f} max {(m[:,2]-t0).max():8d}  end max {(m[:,2]+m[:,3]-t0).max():8d}")

w = t[t[:, 0] == 0]
issue = w[:, 1] & 0xFFFFF; wait = w[:, 1] >> 20
print("weight items: setup+issue %.0f, wait for operands %.0f, mfma+adam+stores %.0f" % (issue.mean(), wait.mean(), (w[:, 3] - issue - wait).mean()))
