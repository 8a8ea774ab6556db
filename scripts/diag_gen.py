"""Per-section shader-clock timeline of gen_kernel (workgroup 0), configs[1] shape."""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from hypad_amd import _C

dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(64, dtype=torch.int64, device=dev)
fn = _C.lib.hypad_diag_set_gen_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
for _ in range(5):
    eng.decoder_iteration(x, idx, None, True)
torch.cuda.synchronize()
t = st.cpu().numpy()
names = ["encoder fwd", "critic_z fwd+bwd", "decoder fwd x2 (trunk)", "head fwd", "critic_x fwd+bwd", "loss + head bwd + dE", "dH1 = dpre W2", "l1 cell bwd + bwd data", "l0 cell bwd + bwd data", "dZ", "encoder bwd"]
d = np.diff(t[:12])
print("total cycles", t[11] - t[0], "=", (t[11] - t[0]) / 2400.0, "us @2.4GHz")
print(", ".join(f"{n} {v}" for n, v in zip(names, d)))
