"""Generator step kernels back to back (hypad_profile_iteration kind 5: 64 launches each) + the graph-replayed epoch, configs[1]."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(1)
step, losses = bench.make_step(eng, x, 1, gen, dev)
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
g, d = [], []
for rep in range(12):
    ms = eng.profile_iteration(5, x, idx, train_mode=True)
    if rep >= 2: g.append(ms[0]); d.append(ms[1])
print("epoch ms min %.4f median %.4f  gen us %.2f  dw us %.2f  finite %s" % (min(ts), float(np.median(ts)), 1e3 * np.mean(g), 1e3 * np.mean(d), bool(torch.isfinite(losses).all())))
