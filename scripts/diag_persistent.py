"""Per-wave shader-clock timeline of one iteration (the sixth) of critic_persistent_kernel, workgroup (chunk 0, signal 0) of each
critic, configs[1] shape.  Needs the development library: python -m hypad_amd.build --dev"""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"
import ctypes
import numpy as np, torch
import bench
from hypad_amd import _C

dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(2 * 32 * 8, dtype=torch.int64, device=dev)
fn = _C.lib.hypad_diag_set_fused_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
perm = torch.stack([torch.randperm(bench.N_WINDOWS, device=dev)[: 6 * bench.B] for _ in range(3)]).to(torch.int32).contiguous()
for _ in range(3):
    eng.train_epoch(x, perm, 6, 2, True)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(2, 32, 8)
names = {0: "loop top", 1: "shares there", 2: "phase A done", 3: "record staged", 4: "fwd0 / phase B", 5: "B seen", 6: "fwd done", 7: "bwd done",
         8: "pre-barrier 1", 9: "barrier 1", 10: "2nd-order | g+dWrf", 11: "barrier 2", 12: "dW gp", 13: "scalars there", 14: "share stored",
         15: "drained+flag", 16: "cleared", 17: "  coef read", 18: "  share tile 0", 19: "  share tile 1", 20: "  share tile 2", 21: "  share tile 3", 22: "  record staged"}
for z, nm in ((0, "critic_x"), (1, "critic_z")):
    t0 = s[z, 0].min()
    print(nm, "iteration cycles (wave 0, top to cleared):", s[z, 16, 0] - s[z, 0, 0])
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 17, 18, 19, 20, 21, 22, 14, 15, 16):
        row = s[z, k]
        print("  %2d %-20s " % (k, names[k]) + " ".join("%6d" % (v - t0) if v else "     -" for v in row))
