"""score_anomalies_sharded at one rank, 125 000 windows: its two pairs of independent chains beside each other or one after the other."""
import sys, time
sys.path.insert(0, ".")
import torch
from hypad_amd import parallel as par
from hypad_amd.models import tadgan
from hypad_amd.utils import anomaly_detection_utils as adu
S, L, n = 100, 20, 125_000
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).cuda().eval(), tadgan.Decoder(S, L, False).cuda().eval(), tadgan.CriticX(S, L).cuda().eval()
x = (torch.rand(n, S, device="cuda") * 2 - 1).contiguous()
real = par._beside
seq = lambda fa, fb, probe: (lambda rb: (fa(), rb))(fb())
calls = {"n": 0}
def first_only(fa, fb, probe):
    calls["n"] += 1
    return real(fa, fb, probe) if calls["n"] % 2 == 1 else seq(fa, fb, probe)
def second_only(fa, fb, probe):
    calls["n"] += 1
    return real(fa, fb, probe) if calls["n"] % 2 == 0 else seq(fa, fb, probe)
def timed(fn, reps=5):
    fn(); best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best
for name, b in (("both beside", real), ("one after the other", seq), ("errors || modes only", first_only), ("z-score || critic score only", second_only), ("both beside", real)):
    par._beside = b; calls["n"] = 0
    t = timed(lambda: par.score_anomalies_sharded(x, enc, dec, cx, S, rec_error_type="dtw", comb="mult", as_tensor=True))
    print("%-32s %.3f ms  %.1f M windows/s" % (name, 1e3 * t, n / t / 1e6))

# ---- the un-sharded pass of bench.py's `scoring` section (hyperbolic forward): whole pass, branches beside / one after the other
from hypad_amd import _C
dec_h = tadgan.Decoder(S, L, True).cuda().eval()
new = lambda *shape: torch.empty(*shape, device="cuda", dtype=torch.float32)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device="cuda")
def forward():
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec_h.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(hyper), _C.ptr(eucl), _C.ptr(hreal),
                                               _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(), ws_bytes, _C.stream()), "fwd")
def numerics():
    true = adu.unroll_true(x)
    pred, _ = adu.unroll_predictions(eucl, False)
    return adu.zscore_clip(adu.rolling_mean(adu._point_wise_error(true, pred), 200)), adu.zscore_clip(adu.rolling_mean(adu._dtw_error(true, pred, 10), 200))
def smoothing():
    return adu._compute_critic_score(adu.kde_modes(critic, S), n // 100)
def seq_pass():
    forward(); numerics(); smoothing()
def seq_pass_kde_first():
    forward(); smoothing(); numerics()
def par_pass():
    forward(); adu.concurrently(numerics, smoothing)
for name, f in (("pass, one after the other", seq_pass), ("pass, critic smoothing first", seq_pass_kde_first), ("pass, branches beside each other", par_pass), ("pass, one after the other", seq_pass)):
    t = timed(f)
    print("%-36s %.3f ms  %.1f M windows/s" % (name, 1e3 * t, n / t / 1e6))
