import sys, time
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
from hypad_amd.models import tadgan
dev = torch.device("cuda", 0)
S, L, n = 100, 20, 125_000
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, True).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
xx = (torch.rand(n, S, device=dev) * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device=dev)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
wsb = _C.lib.hypad_score_workspace_bytes(S, L, 1); ws = torch.empty(wsb // 4, device=dev)
f = lambda: _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(xx), 0, _C.ptr(hyper), _C.ptr(eucl), _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(), wsb, _C.stream()), "fwd")
for rep in range(8):
    ms = bench._event_ms_median(f)
    print("t=%5.1fs  forward %.4f ms" % (time.perf_counter(), ms), flush=True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        for _ in range(50): f()
        torch.cuda.synchronize()
