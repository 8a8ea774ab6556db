"""Per-stage latency of the row-tile kernels (development aid): mean kernel time over back-to-back launches."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
from hypad_amd.models import tadgan

def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us

torch.manual_seed(0)
S, L = 100, 20
enc, dec, cx, cz = tadgan.Encoder(S, L).cuda().eval(), tadgan.Decoder(S, L, True).cuda().eval(), tadgan.CriticX(S, L).cuda().eval(), tadgan.CriticZ(L).cuda().eval()
for rows in (16, 64, 16 * 256, 16 * 2048):
    x = torch.randn(rows, S, device="cuda"); z = torch.randn(rows, L, device="cuda")
    w = torch.randn(384, 128, device="cuda"); xin = torch.randn(rows, 128, device="cuda"); out = torch.empty(rows, 384, device="cuda")
    b = torch.randn(384, device="cuda")
    res = {}
    res["linear128x384"] = timeit(lambda: _C.check(_C.lib.hypad_linear_act_fwd(_C.ptr(xin), _C.ptr(w), _C.ptr(b), _C.ptr(out), rows, 128, 384, 0, _C.stream())))
    w2 = torch.randn(20, 20, device="cuda"); x2 = torch.randn(rows, 20, device="cuda"); o2 = torch.empty(rows, 20, device="cuda"); b2 = torch.randn(20, device="cuda")
    res["linear20x20"] = timeit(lambda: _C.check(_C.lib.hypad_linear_act_fwd(_C.ptr(x2), _C.ptr(w2), _C.ptr(b2), _C.ptr(o2), rows, 20, 20, 0, _C.stream())))
    res["encoder"] = timeit(lambda: enc(x))
    res["decoder"] = timeit(lambda: dec(z))
    res["critic_x"] = timeit(lambda: cx(x))
    res["critic_z"] = timeit(lambda: cz(z))
    u = torch.randn(rows, S, device="cuda") * 0.1; bias = torch.randn(S, device="cuda") * 0.01; o3 = torch.empty_like(u)
    res["head_rows"] = timeit(lambda: _C.check(_C.lib.hypad_mobius_head_fwd(_C.ptr(u), _C.ptr(bias), _C.ptr(o3), rows, S, _C.stream())))
    res["empty(expmap0 1 row)"] = timeit(lambda: _C.check(_C.lib.hypad_expmap0_fwd(_C.ptr(u), _C.ptr(o3), 1, S, _C.stream())))
    print(rows, {k: round(v, 1) for k, v in res.items()})
