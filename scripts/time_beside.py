"""The scoring pass of bench.py's `scoring` section (125 000 windows, hyperbolic forward): the critic smoothing beside the reconstruction
numerics (anomaly_detection_utils.concurrently) or one after the other.  --streams k: after the process has used k other streams."""
import sys, time
sys.path.insert(0, ".")
import torch
from hypad_amd import parallel as par
from hypad_amd.models import tadgan
from hypad_amd.utils import anomaly_detection_utils as adu
S, L, n = 100, 20, 125_000
torch.manual_seed(0)
if "--streams" in sys.argv:          # a process that has used a dozen streams before (the end of a full bench.py run)
    many = [torch.cuda.Stream() for _ in range(int(sys.argv[sys.argv.index("--streams") + 1]))]
    for st in many:
        with torch.cuda.stream(st):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
enc, dec, cx = tadgan.Encoder(S, L).cuda().eval(), tadgan.Decoder(S, L, False).cuda().eval(), tadgan.CriticX(S, L).cuda().eval()
x = (torch.rand(n, S, device="cuda") * 2 - 1).contiguous()
def timed(fn, reps=5):
    fn(); best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best
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
