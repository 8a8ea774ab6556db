"""Decoder forward (16 rows / workgroup) with MFMA-native packed weights vs the streamed + LDS re-shaped gemm_nt."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hypad_amd import _C
from hypad_amd.models import tadgan

torch.manual_seed(0)
S, L, H = 100, 20, 64
dec = tadgan.Decoder(S, L, True).cuda().eval()
sd = {k: v.detach().cpu().numpy() for k, v in dec.state_dict().items()}


def pack(W, rows=None):
    """(N, K) -> blocks [tn][g][lane][4]; rows = list of source rows (gate compaction)."""
    if rows is not None:
        W = W[rows]
    N, K = W.shape
    tn, kg = (N + 15) // 16, (K + 15) // 16
    Wp = np.zeros((tn * 16, kg * 16), np.float32)
    Wp[:N, :K] = W
    blk = Wp.reshape(tn, 16, kg, 4, 4).transpose(0, 2, 3, 1, 4)      # [tn][g][q][j][c]
    return np.ascontiguousarray(blk).reshape(-1)


def gates(h):          # compact [i | g | o] rows of a (4H, in) weight
    return list(range(0, h)) + list(range(2 * h, 4 * h))


parts, offs, cur = [], [], 0
def add(a):
    global cur
    offs.append(cur); parts.append(a.astype(np.float32)); cur += a.size
    pad = (-cur) % 64
    if pad:
        parts.append(np.zeros(pad, np.float32)); cur += pad

add(pack(sd["dense1.weight"])); add(sd["dense1.bias"])
for layer in (0, 1):
    for d in ("", "_reverse"):
        add(pack(sd[f"lstm.weight_ih_l{layer}{d}"], gates(H)))
        add((sd[f"lstm.bias_ih_l{layer}{d}"] + sd[f"lstm.bias_hh_l{layer}{d}"])[gates(H)])
add(pack(sd["dense2.weight"])); add(sd["dense2.bias"])
add(pack(sd["hyperbolic_linear.weight"]))
pk = torch.from_numpy(np.concatenate(parts)).cuda()
offs_c = (ctypes.c_int * 13)(*offs)
head_b = dec.state_dict()["hyperbolic_linear.bias"].contiguous()

fn = _C.lib.hypad_diag_decoder_packed
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
               ctypes.c_void_p, ctypes.c_void_p]
ref_fn = _C.lib.hypad_diag_decoder_timeline
ref_fn.restype = ctypes.c_int
ref_fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
names = ["load z", "dense1", "l0 gates", "l0 cell", "l1 gates", "l1 cell", "dense2", "tanh", "head gemm", "head rows", "store"]
for rows in (16, 64, 16 * 256):
    z = torch.randn(rows, L, device="cuda")
    out_p, out_r = torch.empty(rows, S, device="cuda"), torch.empty(rows, S, device="cuda")
    nblk = (rows + 15) // 16
    for label, call, out in (("packed", lambda st: fn(_C.ptr(pk), offs_c, _C.ptr(head_b), _C.ptr(z), _C.ptr(out_p), rows, S, L, _C.ptr(st), _C.stream()), out_p),
                             ("staged", lambda st: ref_fn(_C.ptr(dec.arena()), _C.ptr(z), _C.ptr(out_r), rows, S, L, 1, 512, _C.ptr(st), _C.stream()), out_r)):
        st = torch.zeros(nblk, 64, dtype=torch.int64, device="cuda")
        for rep in range(5):
            assert call(st) == 0
        torch.cuda.synchronize()
        s = st.cpu().numpy()[:, 24:48].reshape(nblk, 12, 2)
        cyc = np.diff(s[:, :, 0], axis=1)
        tot = s[:, -1, 0] - s[:, 0, 0]
        print(f"{label:7s} rows={rows:5d}: total {np.median(tot)} cycles; " + ", ".join(f"{n} {np.median(cyc[:, i]):.0f}" for i, n in enumerate(names)))
    ref = dec(z.view(1, rows, L))[0].reshape(rows, S)
    print("   max |packed - module| =", float((out_p - ref).abs().max()), " max |staged - module| =", float((out_r - ref).abs().max()))
