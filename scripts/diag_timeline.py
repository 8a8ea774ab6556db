import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from hypad_amd import _C
from hypad_amd.models import tadgan
torch.manual_seed(0)
S, L = 100, 20
dec = tadgan.Decoder(S, L, True).cuda().eval()
fn = _C.lib.hypad_diag_decoder_timeline
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
names = ["load z", "dense1", "l0 gates", "l0 cell", "l1 gates", "l1 cell", "dense2", "tanh", "head gemm", "head rows", "store"]
for mt, threads, rows in ((1, 512, 16), (1, 1024, 16), (2, 512, 32), (1, 512, 16 * 256)):
    z = torch.randn(rows, L, device="cuda"); hyper = torch.empty(rows, S, device="cuda")
    nblk = (rows + mt * 16 - 1) // (mt * 16)
    st = torch.zeros(nblk, 64, dtype=torch.int64, device="cuda")
    for rep in range(5):
        rc = fn(_C.ptr(dec.arena()), _C.ptr(z), _C.ptr(hyper), rows, S, L, mt, threads, _C.ptr(st), _C.stream())
        assert rc == 0, rc
    torch.cuda.synchronize()
    for rep in range(2):
        s = st.cpu().numpy()[:, 24 * rep:24 * rep + 24].reshape(nblk, 12, 2)
        cyc = np.diff(s[:, :, 0], axis=1); wall = np.diff(s[:, :, 1], axis=1)
        tot_c = s[:, -1, 0] - s[:, 0, 0]; tot_w = s[:, -1, 1] - s[:, 0, 1]
        print(f"MT={mt} threads={threads} rows={rows} pass{rep}: total {np.median(tot_w)/100:.1f} us, clock {np.median(tot_c)/np.median(tot_w)*100:.0f} MHz")
        print("   " + ", ".join(f"{n} {np.median(cyc[:, i]):.0f}c" for i, n in enumerate(names)))
ref = dec(torch.randn(16, L, device="cuda"))
