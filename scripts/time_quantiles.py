"""hypad_quantiles / hypad_critic_score against torch.quantile (HIP events, back-to-back calls), 125 099 and 1 000 099 fp64 values."""
import sys
sys.path.insert(0, ".")
import torch
from hypad_amd.utils import anomaly_detection_utils as adu
def ev(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(30_000_000)          # ~12 ms: the host enqueues everything behind it, the events see GPU time only
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for n in (125_099, 1_000_099):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, device="cuda", generator=g).double()
    q = torch.tensor([0.25, 0.75], dtype=torch.float64, device="cuda")
    print(n, "quantiles us %.1f" % ev(lambda: adu.quantiles(x, (0.25, 0.75))), "torch.quantile us %.1f" % ev(lambda: torch.quantile(x, q)),
          "critic score (incl. rolling mean) us %.1f" % ev(lambda: adu._compute_critic_score(x, n // 100)))
