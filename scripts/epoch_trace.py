"""Runs a few hoisted epochs (configs[1] shape) -- target of rocprofv3 --kernel-trace."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench

dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(1)
nb, nc, B = bench.N_BATCHES, bench.N_CRITICS, bench.B
for rep in range(4):
    perm = torch.rand(nc + 1, bench.N_WINDOWS, device=dev, generator=gen).argsort(dim=1)[:, : nb * B].to(torch.int32).contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_epoch(x, perm, nb, nc, True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("epoch ms %.2f (enqueue %.2f)" % ((time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3), flush=True)
