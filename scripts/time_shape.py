"""ms per captured epoch of one model at a given shape: time_shape.py S B N [reps]   (A/B runs: scripts/ab_variants.sh)"""
import sys, time
sys.path.insert(0, ".")
import torch, bench
S, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
cfg = bench.Cfg("x", S=S, B=B, n_windows=N, data="uniform")
eng, x = bench.build_engine(1, 0, True, dev, cfg)
step, losses = bench.make_step(eng, x, 1, torch.Generator(device=dev).manual_seed(1), dev, graph=True, cfg=cfg)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): step()
torch.cuda.synchronize()
print("S %d B %d N %d epoch ms %.3f" % (S, B, N, (time.perf_counter() - t0) / reps * 1e3), "finite", bool(torch.isfinite(losses).all()), "status", eng.status())
