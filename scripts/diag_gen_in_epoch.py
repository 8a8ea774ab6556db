"""The generator kernel's per-chain shader-clock timeline INSIDE an epoch (the last generator launch of an eager epoch; development library):
what scripts/diag_gen.py reports for a stand-alone iteration, measured where the epoch's steps run back to back."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"
import ctypes
import numpy as np, torch
import bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(3 * 48 * 8 + 64 + 2 * 1024 + 64, dtype=torch.int64, device=dev)      # (the dW kernel stamps behind the chains)
fn = _C.lib.hypad_diag_set_gen_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
nb, nc = 29, 5
perm = torch.stack([torch.randperm(bench.N_WINDOWS, device=dev)[: nb * bench.B] for _ in range(nc + 1)]).to(torch.int32).contiguous()
for _ in range(3):
    eng.train_epoch(x, perm, nb, nc, True)
torch.cuda.synchronize()
tw = st[: 3 * 48 * 8].cpu().numpy().reshape(3, 48, 8)
for role, nm in ((0, "G"), (1, "R"), (2, "Z")):
    t = tw[role]
    marks = [k for k in range(48) if t[k].max() > 0]
    if not marks:
        print("role", nm, "no stamps"); continue
    first, last = min(t[k][t[k] > 0].min() for k in marks), max(t[k].max() for k in marks)
    print(f"role {nm}: first stamp -> last stamp {last - first} cycles; marks {marks[:6]}..{marks[-3:]}")
    r = t[:, 0]
    ks = [k for k in (0, 14, 15, 12, 13, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11) if r[k] > 0]
    print("   wave 0: " + " ".join(f"{k}:{r[k] - r[ks[0]]}" for k in ks))
base = min(tw[role][k][tw[role][k] > 0].min() for role in range(3) for k in range(48) if tw[role][k].max() > 0)
for role, nm in ((0, "G"), (1, "R"), (2, "Z")):
    t = tw[role]
    ks = [k for k in range(48) if t[k].max() > 0]
    if ks:
        print(f"role {nm}: starts at {min(t[k][t[k] > 0].min() for k in ks) - base}, ends at {max(t[k].max() for k in ks) - base} (cycles since the launch's first stamp)")
