"""GPU time of Engine.critic_x_iteration with z / alpha in device memory vs read from pinned host memory through its own address."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
xb = x[:, :64].contiguous()
zd = torch.randn(1, 64, 20, device=dev); ad = torch.rand(1, 64, 100, device=dev)
zh = zd.cpu().pin_memory(); ah = ad.cpu().pin_memory()
for name, (z, a) in (("device", (zd, ad)), ("pinned host", (zh, ah))):
    for _ in range(5): eng.critic_x_iteration(xb, None, z, a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for _ in range(100): eng.critic_x_iteration(xb, None, z, a)
    e1.record(); torch.cuda.synchronize()
    print(name, "critic_x GPU us per call %.1f" % (e0.elapsed_time(e1) / 100 * 1e3))
