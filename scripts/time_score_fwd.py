"""hypad_score_forward_packed at 125 000 windows (hyperbolic): ms per call by HIP events (A/B of library builds: HYPAD_LIB_PATH)."""
import os, sys
sys.path.insert(0, ".")
import torch
from hypad_amd import _C
from hypad_amd.models import tadgan
S, L, n = 100, 20, 125_000
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).cuda().eval(), tadgan.Decoder(S, L, True).cuda().eval(), tadgan.CriticX(S, L).cuda().eval()
x = (torch.rand(n, S, device="cuda") * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device="cuda")
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
wb = _C.lib.hypad_score_workspace_bytes(S, L, 1); ws = torch.empty(wb // 4, device="cuda")
def fwd():
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(hyper), _C.ptr(eucl), _C.ptr(hreal),
                                               _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(), wb, _C.stream()), "fwd")
for _ in range(3): fwd()
best = 1e9
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fwd()
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 10)
print("%s: forward %.4f ms  checksum %.6f" % (os.path.basename(os.environ.get("HYPAD_LIB_PATH", "product")), best, float(hyper.double().sum() + critic.double().sum())))
