"""Why does bench.py's `secondary` graph replay come out slower than its eager run?  Replays bench.main's order piece by piece."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
gen = torch.Generator(device=dev).manual_seed(100)
def sig8(tag):
    o = bench.bench_signals(8, 0, dev, gen)
    print(tag, "graph %.3f eager %.3f" % (o["graph_ms_per_step"], o["eager_ms_per_step"]), flush=True)
sig8("alone")
eng, x = bench.build_engine(1, 0, True, dev)
step, losses = bench.make_step(eng, x, 1, gen, dev, graph=True)
for _ in range(23): step()
torch.cuda.synchronize()
sig8("after a 1-signal graph engine ran")
prof = bench.profile_kernels(eng, x, 1, dev)
sig8("after profile_kernels")
del eng, x, step
sig8("after the engine is gone")
