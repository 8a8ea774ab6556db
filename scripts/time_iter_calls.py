"""Engine.critic_x_iteration / critic_z_iteration / decoder_iteration: GPU time per call (HIP events behind a long sleep kernel: the host is
enqueued ahead) and host time per call (wall clock of the enqueue loop), configs[1].  HYPAD_ITER_PHASE=0: stand-alone critic launches."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
xb = x[:, :64].contiguous()
z = torch.randn(1, 64, 20, device=dev); ax = torch.rand(1, 64, 100, device=dev); az = torch.rand(1, 64, 20, device=dev)
calls = {"critic_x": lambda: eng.critic_x_iteration(xb, None, z, ax), "critic_z": lambda: eng.critic_z_iteration(xb, None, z, az),
         "decoder": lambda: eng.decoder_iteration(xb, None, z)}
for name, fn in calls.items():
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(40_000_000)
    a.record()
    t0 = time.perf_counter()
    for _ in range(100): fn()
    host = (time.perf_counter() - t0) / 100 * 1e6
    b.record(); torch.cuda.synchronize()
    print(name, "GPU %.1f us per call, host enqueue %.1f us per call" % (a.elapsed_time(b) / 100 * 1e3, host))
