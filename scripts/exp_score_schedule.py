"""Two schedules of one scoring pass (125 000 windows): the critic branch (critic value -> KDE modes -> trimmed z-score -> rolling mean) behind
the fused forward (as score_anomalies queues it today) or BESIDE it from the start (its only input is the windows)."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
from hypad_amd.models import tadgan
from hypad_amd.utils import anomaly_detection_utils as adu
dev = torch.device("cuda", 0)
S, L, n, smooth = 100, 20, 125_000, 200
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, True).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
x = (torch.rand(n, S, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device=dev)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
wsb = _C.lib.hypad_score_workspace_bytes(S, L, 1); ws, ws2 = torch.empty(wsb // 4, device=dev), torch.empty(wsb // 4, device=dev)
def fwd(h, e, hr, c, d, w):
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(h), _C.ptr(e), _C.ptr(hr), _C.ptr(c), _C.ptr(d),
                                               n, S, L, 1, w.data_ptr(), wsb, _C.stream()), "fwd")
def numerics():
    true = adu.unroll_true(x)
    pred, _ = adu.unroll_predictions(eucl, False)
    e1 = adu.rolling_mean(adu._point_wise_error(true, pred), smooth)
    e2 = adu.rolling_mean(adu._dtw_error(true, pred, 10), smooth)
    return adu.zscore_clip(e1), adu.zscore_clip(e2)
smoothing = lambda: adu._compute_critic_score(adu.kde_modes(critic, S), n // 100)
def behind():
    fwd(hyper, eucl, hreal, critic, dist, ws)
    return adu.concurrently(numerics, smoothing)
def beside():
    def main():
        fwd(hyper, eucl, hreal, None, dist, ws)
        return numerics()
    def side():
        fwd(None, None, None, critic, None, ws2)
        return smoothing()
    return adu.concurrently(main, side)
def timed(fn, reps=5):
    fn(); best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best
ra = behind(); torch.cuda.synchronize(); a = [t.clone() for t in (*ra[0], ra[1])]
rb = beside(); torch.cuda.synchronize(); b = [t.clone() for t in (*rb[0], rb[1])]
print("same results:", all(torch.equal(p, q) or bool(((p == q) | (p.isnan() & q.isnan())).all()) for p, q in zip(a, b)))
for name, f in (("critic branch behind the forward", behind), ("critic branch beside the forward", beside), ("behind again", behind), ("beside again", beside)):
    t = timed(f)
    print("%-36s %.4f ms per pass  %.1f M windows/s" % (name, 1e3 * t, n / t / 1e6))
