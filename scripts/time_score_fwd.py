"""What-if timings of the fused scoring forward (125 000 windows): which outputs it writes, with / without the critic inside, the critic alone."""
import sys
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
from hypad_amd.models import tadgan
dev = torch.device("cuda", 0)
S, L, n = 100, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, True).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
xx = (torch.rand(n, S, device=dev) * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device=dev)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
wsb = _C.lib.hypad_score_workspace_bytes(S, L, 1); ws = torch.empty(wsb // 4, device=dev)
def fwd(h, e, hr, c, d):
    return lambda: _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(xx), 0, _C.ptr(h) if h is not None else None,
                                                              _C.ptr(e) if e is not None else None, _C.ptr(hr) if hr is not None else None, _C.ptr(c) if c is not None else None,
                                                              _C.ptr(d) if d is not None else None, n, S, L, 1, ws.data_ptr(), wsb, _C.stream()), "fwd")
for name, f in (("all outputs (bench)", fwd(hyper, eucl, hreal, critic, dist)), ("no critic", fwd(hyper, eucl, hreal, None, dist)),
                ("eucl + critic + dist (the scoring pass)", fwd(None, eucl, None, critic, dist)), ("dist only", fwd(None, None, None, None, dist))):
    ms = bench._event_ms_median(f)
    print("%-42s %.4f ms  %.1f M windows/s  %.1f %% of fp32 MFMA peak" % (name, ms, n / ms / 1e3, 340312.0 * n / (ms * 1e-3) / 157.3e12 * 100))
out = new(n)
ms = bench._event_ms_median(lambda: _C.check(_C.lib.hypad_critic_x_fwd(_C.ptr(cx.arena()), _C.ptr(xx), _C.ptr(out), n, S, L, None, _C.stream()), "cx"))
print("hypad_critic_x_fwd alone                   %.4f ms" % ms)
# the numerics kernels behind the forward (bench.py roofline_scoring times the same calls)
crit = torch.randn(n, device=dev)
modes = torch.empty(n + S - 1, dtype=torch.float64, device=dev)
ms = bench._event_ms_median(lambda: _C.check(_C.lib.hypad_kde_mode(_C.ptr(crit), _C.ptr(modes), n, S, _C.stream()), "kde"))
print("kde_mode (random-normal critic values)     %.4f ms" % ms)
_forward_all = fwd(hyper, eucl, hreal, critic, dist); _forward_all()
ms = bench._event_ms_median(lambda: _C.check(_C.lib.hypad_kde_mode(_C.ptr(critic), _C.ptr(modes), n, S, _C.stream()), "kde"))
print("kde_mode (this model's critic values)      %.4f ms" % ms)
pred32 = torch.empty(n + S - 1, device=dev)
ms = bench._event_ms_median(lambda: _C.check(_C.lib.hypad_unroll_median(_C.ptr(eucl), _C.ptr(pred32), None, n, S, _C.stream()), "unroll"))
print("unroll_median                              %.4f ms" % ms)
