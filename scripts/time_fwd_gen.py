"""A/B target (scripts/ab_libs.sh run time_fwd_gen.py): ms per captured configs[1] epoch, us per generator / dW launch (HIP events), and
ms per fused scoring forward of 125 000 windows -- one line."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, bench
from hypad_amd import _C
from hypad_amd.models import tadgan
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(1)
step, losses = bench.make_step(eng, x, 1, gen, dev)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(60): step()
torch.cuda.synchronize(); ep = (time.perf_counter() - t0) / 60 * 1e3
prof = bench.profile_kernels(eng, x, 1, dev, reps=10)
S, L, n = 100, 20, 125_000
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, True).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
xx = (torch.rand(n, S, device=dev) * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device=dev)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
wsb = _C.lib.hypad_score_workspace_bytes(S, L, 1); ws = torch.empty(wsb // 4, device=dev)
fwd = lambda: _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(xx), 0, _C.ptr(hyper), _C.ptr(eucl),
                                                        _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(), wsb, _C.stream()), "fwd")
ms = bench._event_ms_median(fwd)
print("epoch ms %.3f  gen us %.2f  dW us %.2f  critic it us %.3f  scoring forward ms %.4f" % (ep, 1e3 * prof["kern_ms"]["gen"], 1e3 * prof["kern_ms"]["dw_gen"], 1e3 * prof["kern_ms"]["critic_iteration"], ms))
