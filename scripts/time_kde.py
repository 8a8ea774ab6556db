"""kde_mode_kernel / unroll_median_kernel launch times (HIP events), 125 000 windows of 100, random-normal critic values."""
import sys
sys.path.insert(0, ".")
import torch
from hypad_amd import _C
dev = "cuda"
n, S = 125_000, 100
g = torch.Generator(device=dev).manual_seed(0)
critic = torch.randn(n, device=dev, generator=g)
modes = torch.empty(n + S - 1, device=dev, dtype=torch.float64)
yh = torch.randn(n, S, device=dev, generator=g)
med = torch.empty(n + S - 1, device=dev)
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
k = ev(lambda: _C.lib.hypad_kde_mode(_C.ptr(critic), _C.ptr(modes), n, S, _C.stream()))
u = ev(lambda: _C.lib.hypad_unroll_median(_C.ptr(yh), _C.ptr(med), None, n, S, _C.stream()))
print("kde us %.1f" % k, "unroll us %.1f" % u, "checksum %.6f" % float(modes.sum()))
