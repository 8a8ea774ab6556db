"""time_graph.py --graph-only plus a checksum of the losses and weights after the timed epochs (A/B builds that must give the same bits)."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng, x = bench.build_engine(1, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(1)
losses = torch.empty(1, 11 * bench.N_BATCHES, 4, device=dev)
buf = torch.empty(6, bench.N_BATCHES * bench.B, dtype=torch.int32, device=dev)
def step():
    perm = torch.rand(6, bench.N_WINDOWS, device=dev, generator=gen).argsort(dim=1)[:, : bench.N_BATCHES * bench.B]
    buf.copy_(perm)
    eng.train_epoch_graph(x, buf, bench.N_BATCHES, 5, True, losses=losses)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
ck = float(losses.double().sum()) , float(sum(eng.params[k].double().abs().sum() for k in ("enc", "dec", "cx", "cz")))
print("graph epoch ms %.3f" % (dt * 1e3), "finite", bool(torch.isfinite(losses).all()), "checksum %.9f %.9f" % ck, "status", eng.status())
