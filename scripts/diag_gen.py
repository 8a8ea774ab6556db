"""Per-section shader-clock timeline of gen_kernel (workgroup 0), configs[1] shape."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from hypad_amd import _C

dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(3 * 48 * 8, dtype=torch.int64, device=dev)
fn = _C.lib.hypad_diag_set_gen_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
for _ in range(5):
    eng.decoder_iteration(x, idx, None, True)
torch.cuda.synchronize()
tw = st.cpu().numpy().reshape(3, 48, 8)        # [role][mark][wave]
t = tw[:, :, 0]
# interval that STARTS at mark k (marks 0-11 of gen_body, in program order)
names = ["start -> encoder done (R) / z + critic staging (G)", "(encoder done -> trunk start)", "decoder trunk", "head forward",
         "critic_x fwd+bwd (G)", "loss + head backward + dE (tanh' in its epilogue)", "dH1 + layer-1 cell backward",
         "layer-1 backward product + layer-0 cell backward", "layer-0 backward product", "dZ", "encoder backward", ""]
for role, nm in ((0, "G"), (1, "R")):
    r = t[role]
    print(f"role {nm}: total {r[11] - r[0]} cycles = {(r[11] - r[0]) / 2400.0:.1f} us")
    ks = [k for k in range(12) if r[k] > 0]
    print("   " + ", ".join(f"{names[k1]} {r[k2] - r[k1]}" for k1, k2 in zip(ks[:-1], ks[1:])))
    sub = {"setup+warm": (0, 14), "x gather": (14, 15), "x gather wait": (12, 13), "enc layer+dense": (13, 1), "head gemm": (20, 21), "head rows": (22, 4),
           "rowdist": (5, 23), "head bwd rows": (23, 24), "bias-gradient column sums": (24, 25), "dE gemm": (25, 26)}
    print("   sub: " + ", ".join(f"{k} {r[b] - r[a]}" for k, (a, b) in sub.items() if r[a] > 0 and r[b] > 0))

# per-wave arrival at every mark, relative to wave 0's start: who is late for the barrier that follows
order = [0, 14, 15, 12, 13, 1, 2, 16, 17, 18, 19, 27, 28, 40, 41, 42, 29, 30, 31, 3, 20, 21, 22, 4, 5, 23, 24, 25, 26, 6, 7, 32, 33, 34, 35, 36, 8, 37, 38, 39, 9, 10, 11]
for role, nm in ((0, "G"), (1, "R")):
    print(f"role {nm}: mark: arrival of waves 0..7 (cycles since start), spread")
    base = tw[role, 0, 0]
    for k in order:
        r = tw[role, k]
        if r.max() > 0:
            rel = r - base
            print(f"   mark {k:2d}: " + " ".join(f"{int(v):7d}" for v in rel) + f"   spread {int(rel.max() - rel.min())}")
