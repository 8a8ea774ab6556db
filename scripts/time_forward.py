"""score_forward_packed_kernel launch time (HIP events), 125 000 windows of 100; `--eucl` for the Euclidean model; checksum of the outputs.
With `--epoch32`: also one 32-signal training epoch (graph replay), whose record precompute launch uses the same tile functions."""
import sys
sys.path.insert(0, ".")
import torch
from hypad_amd import _C
from hypad_amd.models import tadgan

dev = torch.device("cuda", 0)
S, L, n = 100, 20, 125_000
hyp = "--eucl" not in sys.argv
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, hyp).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.rand(n, S, device=dev, generator=g) * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)


def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def full():
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(hyper) if hyp else None,
                                               _C.ptr(eucl), _C.ptr(hreal) if hyp else None, _C.ptr(critic), _C.ptr(dist) if hyp else None, n, S, L, int(hyp),
                                               ws.data_ptr(), ws_bytes, _C.stream()), "score_forward")


def lean():          # what the scorers ask for: critic + row distance (hyperbolic) or critic + reconstruction (Euclidean)
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, None,
                                               None if hyp else _C.ptr(eucl), None, _C.ptr(critic), _C.ptr(dist) if hyp else None, n, S, L, int(hyp),
                                               ws.data_ptr(), ws_bytes, _C.stream()), "score_forward")


t_full, t_lean = ev(full), ev(lean)
full(); torch.cuda.synchronize()
cs = float(critic.double().sum() + (dist.double().sum() if hyp else eucl.double().sum()))
msg = "forward us full %.1f lean %.1f checksum %.6f" % (t_full, t_lean, cs)
if "--epoch32" in sys.argv:
    import bench
    eng, xw = bench.build_engine(32, 0, True, dev)
    gen = torch.Generator(device=dev).manual_seed(100)
    step, losses = bench.make_step(eng, xw, 32, gen, dev, graph=True)
    for _ in range(3):
        step()
    msg += " epoch32 ms %.3f" % (ev(step, 10) / 1e3)
print(msg)
