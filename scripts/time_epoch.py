"""Epoch time of configs[1] (and --spg signals per GPU) + per-kernel HIP-event times; prints which critic-phase form ran."""
import sys, time
sys.path.insert(0, ".")
import argparse
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--spg", type=int, default=1)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(args.spg, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(1)
losses = torch.empty(args.spg, 11 * bench.N_BATCHES, 4, device=dev)

def step():
    perm = torch.rand(6, bench.N_WINDOWS, device=dev, generator=gen).argsort(dim=1)[:, : bench.N_BATCHES * bench.B]
    eng.train_epoch(x, perm.to(torch.int32).contiguous(), bench.N_BATCHES, 5, True, losses=losses)

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print("persistent:", eng.critic_phase_persistent(), "epoch ms: %.3f" % (dt * 1e3), "windows/s: %.0f" % (args.spg * bench.N_BATCHES * bench.B / dt),
      "finite:", bool(torch.isfinite(losses).all()), "mean losses", losses.mean(dim=(0, 1)).tolist())
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
acc = {4: [], 2: []}
for rep in range(30):
    for kind in (4, 2):
        ms = eng.profile_iteration(kind, x, idx, train_mode=True)
        if rep >= 5:
            acc[kind].append(ms)
import numpy as np
print("kind4 [precompute(29 its), first/reinit, per-iteration] us:", (np.mean(acc[4], 0) * 1e3).round(2).tolist())
print("kind2 [gen, dw] us:", (np.mean(acc[2], 0) * 1e3).round(2).tolist())
