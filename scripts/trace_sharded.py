"""Target for `rocprofv3 --kernel-trace`: the two sharded scorers on a one-rank RCCL group, 125 000 windows, six calls each
(scripts/trace_sharded.sh prints one call's kernel timeline)."""
import os, socket, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypad_amd import parallel as par
from hypad_amd.models import tadgan
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
S, L, n = 100, 20, 125_000
with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev).manual_seed(3)
series = (torch.rand(n + S - 1, device=dev, generator=g) * 2 - 1).contiguous()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for hyperbolic in (True, False):
    if which == "hyper" and not hyperbolic or which == "dtw" and hyperbolic:
        continue
    torch.manual_seed(7)
    enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, hyperbolic).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
    if hyperbolic:
        fn = lambda: par.score_windows_sharded(series, enc, dec, cx, S, "mult", x_row_stride=1, as_tensor=True)
    else:
        y = series.unfold(0, S, 1)[:n].contiguous()
        fn = lambda: par.score_anomalies_sharded(y, enc, dec, cx, S, rec_error_type="dtw", comb="mult", as_tensor=True)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        torch.cuda._sleep(2_000_000)          # ~1 ms marker gap between the calls in the trace
        fn()
    torch.cuda.synchronize()
    print("hyperbolic" if hyperbolic else "dtw", "ms per call incl. marker", (time.perf_counter() - t0) / 6 * 1e3)
dist.destroy_process_group()
