"""Per-kernel timing of the scoring numerics (125 000 windows x 100)."""
import sys, time
import torch
sys.path.insert(0, ".")
from hypad_amd.utils import anomaly_detection_utils as adu

dev = "cuda"
n, S = 125_000, 100
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(n, S, device=dev, generator=g) * 2 - 1).contiguous()
yh = (x + 0.05 * torch.randn(n, S, device=dev, generator=g)).contiguous()

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

true = adu.unroll_true(x)
pred, _ = adu.unroll_predictions(yh, False)
e = adu._point_wise_error(true, pred)
print("unroll_true        %8.1f us" % timed(lambda: adu.unroll_true(x)))
print("unroll median      %8.1f us" % timed(lambda: adu.unroll_predictions(yh, False)))
print("unroll median+5num %8.1f us" % timed(lambda: adu.unroll_predictions(yh, True)))
print("point error        %8.1f us" % timed(lambda: adu._point_wise_error(true, pred)))
print("dtw error          %8.1f us" % timed(lambda: adu._dtw_error(true, pred, 10)))
print("area error         %8.1f us" % timed(lambda: adu._area_error(true, pred, 10)))
print("rolling mean 200   %8.1f us" % timed(lambda: adu.rolling_mean(e, 200)))
print("zscore clip        %8.1f us" % timed(lambda: adu.zscore_clip(e)))
print("kde modes (critic) %8.1f us" % timed(lambda: adu.kde_modes(torch.randn(n, device=dev), S), reps=3))
