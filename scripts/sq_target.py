"""Target of scripts/sq_profile.sh: a few launches of the throughput kernels (score forward as the scorers call it, KDE modes, un-roll median,
a 32-signal epoch for critic_phase_precompute_kernel).  Run directly after `--`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypad_amd import _C  # noqa: E402
from hypad_amd.models import tadgan  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
S, L, n = 100, 20, 125_000
reps = int(os.environ.get("REPS", "3"))
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, True).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.rand(n, S, device=dev, generator=g) * 2 - 1).contiguous()
critic, dist = torch.empty(n, device=dev), torch.empty(n, device=dev)
ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
modes = torch.empty(n + S - 1, device=dev, dtype=torch.float64)
yh = torch.randn(n, S, device=dev, generator=g)
med = torch.empty(n + S - 1, device=dev)
cv = torch.randn(n, device=dev, generator=g)
for _ in range(reps):
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, None, None, None, _C.ptr(critic),
                                               _C.ptr(dist), n, S, L, 1, ws.data_ptr(), ws_bytes, _C.stream()), "score_forward")
    _C.check(_C.lib.hypad_kde_mode(_C.ptr(cv), _C.ptr(modes), n, S, _C.stream()), "kde")
    _C.check(_C.lib.hypad_unroll_median(_C.ptr(yh), _C.ptr(med), None, n, S, _C.stream()), "unroll")
if os.environ.get("EPOCH32", "1") == "1":
    import bench
    eng, xw = bench.build_engine(32, 0, True, dev)
    gen = torch.Generator(device=dev).manual_seed(100)
    step, losses = bench.make_step(eng, xw, 32, gen, dev, graph=False)
    for _ in range(reps):
        step()
torch.cuda.synchronize()
print("sq target done")
