"""Per-stage shader-clock timeline of critic_fused_pair_kernel (first chunk), configs[1] shape."""
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
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
for _ in range(5):
    ms = eng.profile_iteration(4, x, idx, True)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(2, 64)
print("events ms", ms)
for z, nm in ((0, "critic_x"), (1, "critic_z")):
    nh = 4 if z == 0 else 2
    names = ["zero+stage", "P0 rows+interp", "P0 prefetch+masks+sync"] + [f"fwd{l}" for l in range(nh)] + [f"bwd{l}" for l in range(nh - 2, -1, -1)] + ["g"] + [f"ep{l}" for l in range(nh)] + ["dbias", "dW+sync"]
    t = s[z]
    n = len(names)
    d = np.diff(t[: n + 1])
    print(nm, "total cycles", t[41] - t[0], "first chunk", t[n] - t[1], "rest(3 chunks)", t[40] - t[n], "tail", t[41] - t[40])
    print("   " + ", ".join(f"{a} {b}" for a, b in zip(names, d)))
