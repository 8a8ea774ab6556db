"""unroll_median / kde_mode kernel timings alone (125 000 windows of 100): A/B runs with scripts/ab_variants.sh."""
import sys
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
S, n = 100, 125_000
g = torch.Generator(device=dev).manual_seed(0)
eucl = (torch.rand(n, S, device=dev, generator=g) * 2 - 1).contiguous()
pred32 = torch.empty(n + S - 1, device=dev)
crit = torch.randn(n, device=dev, generator=g)
modes = torch.empty(n + S - 1, dtype=torch.float64, device=dev)
mu = bench._event_ms_median(lambda: _C.check(_C.lib.hypad_unroll_median(_C.ptr(eucl), _C.ptr(pred32), None, n, S, _C.stream()), "unroll"))
mk = bench._event_ms_median(lambda: _C.check(_C.lib.hypad_kde_mode(_C.ptr(crit), _C.ptr(modes), n, S, _C.stream()), "kde"))
print("unroll_median %.4f ms | kde_mode %.4f ms | checksum %.6f %.6f" % (mu, mk, float(pred32.double().sum()), float(modes.sum())))
