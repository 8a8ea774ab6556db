import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
for spg in (8, 16, 32):
    eng, x = bench.build_engine(spg, 0, True, dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    step, losses = bench.make_step(eng, x, spg, gen, dev)
    for _ in range(3): step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    print("%d signals: %.3f ms per epoch, status %d" % (spg, best, eng.status()))
