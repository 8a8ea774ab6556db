import sys, io, os, time, tempfile
sys.path.insert(0, ".")
import torch
from hypad_amd.models import tadgan
from hypad_amd.train import _SavedLayout
d = tempfile.mkdtemp()
for name, m in (("decoder", tadgan.Decoder(100, 20, True).cuda()), ("encoder", tadgan.Encoder(100, 20).cuda()), ("critic_x", tadgan.CriticX(100, 20).cuda())):
    f = os.path.join(d, name + ".pt")
    t0 = time.perf_counter()
    for _ in range(50): torch.save(m, f)
    t1 = time.perf_counter()
    buf = io.BytesIO(); torch.save(m, buf); raw = buf.getvalue()
    lay = _SavedLayout.parse(raw, m.arena().detach().reshape(-1).cpu().numpy().tobytes())
    src = m.arena().detach().clone()
    t2 = time.perf_counter()
    for _ in range(50): lay.write(f, src.reshape(-1).cpu().numpy().tobytes())
    t3 = time.perf_counter()
    for _ in range(50): b = src.reshape(-1).cpu().numpy().tobytes()
    t4 = time.perf_counter()
    print("%-9s %7d bytes: torch.save %.3f ms, layout.write %.3f ms (of which D2H + tobytes %.3f ms)" % (name, len(raw), 1e3 * (t1 - t0) / 50, 1e3 * (t3 - t2) / 50, 1e3 * (t4 - t3) / 50))
