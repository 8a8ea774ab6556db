"""Four-row tile product (v_mfma_f32_4x4x1, tile_gemm.h gemm4_nt_packed_epi) against the sixteen-row one over the same packed weights:
results of rows 0..3 and shader-clock cycles per call of one 512-thread workgroup (development library)."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"
import ctypes
import numpy as np, torch
from hypad_amd import _C


def pack(W):
    N, K = W.shape
    tn, kg = (N + 15) // 16, (K + 15) // 16
    Wp = np.zeros((tn * 16, kg * 16), np.float32)
    Wp[:N, :K] = W
    return np.ascontiguousarray(Wp.reshape(tn, 16, kg, 4, 4).transpose(0, 2, 3, 1, 4)).reshape(-1)      # [tn][g][q][j][c]


fn = _C.lib.hypad_diag_gemm4
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rng = np.random.default_rng(0)
for K, N in ((128, 384), (100, 300), (384, 128), (100, 100), (20, 50), (50, 384)):
    W = rng.standard_normal((N, K)).astype(np.float32) / np.sqrt(K)
    X = rng.standard_normal((16, K)).astype(np.float32)
    Wp, Xd = torch.from_numpy(pack(W)).cuda(), torch.from_numpy(X).cuda()
    y16, y4, out = torch.zeros(16, N, device="cuda"), torch.zeros(4, N, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda")
    rc = fn(Wp.data_ptr(), Xd.data_ptr(), K, N, y16.data_ptr(), y4.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    e16, e4 = np.abs(y16.cpu().numpy() - ref).max(), np.abs(y4.cpu().numpy() - ref[:4]).max()
    c = out.cpu().tolist()
    print(f"K {K:3d} N {N:3d}: rc {rc}  max err 16-row {e16:.2e}  4-row {e4:.2e}   cycles 16-row {c[0]} / {c[1]}   4-row {c[2]} / {c[3]}")
