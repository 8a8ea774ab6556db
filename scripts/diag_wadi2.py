"""Round 6 diagnosis: bench_signals(WADI) fails with non-finite losses where per-epoch-synchronised loops do not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

dev = torch.device("cuda", 0)
for S, sync_every in ((123, 0), (123, 1), (124, 0), (51, 0), (123, 0)):
    gen = torch.Generator(device=dev).manual_seed(100)
    cfg = bench.Cfg("x", S=S, B=64, n_windows=20480, data="uniform")
    eng, x = bench.build_engine(1, 0, True, dev, cfg)
    step, losses = bench.make_step(eng, x, 1, gen, dev, graph=True, cfg=cfg)
    keep = []
    for ep in range(16):
        step()
        keep.append(losses.clone())          # (device-side copy on the same stream: no host synchronisation)
        if sync_every:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    first = None
    nb = cfg.nb
    for ep, l in enumerate(keep):
        l = l.cpu().numpy()[0]
        bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
        if len(bad):
            first = (ep, int(bad[0]), len(bad), "gen" if bad[0] >= 10 * nb else ("cx" if bad[0] % 2 == 0 else "cz"), l[max(0, bad[0] - 1): bad[0] + 2].tolist())
            break
    print("S", S, "sync_every", sync_every, "status", eng.status(), "first non-finite (epoch, row, count, kind, rows):", first, flush=True)
    del eng, x, step, losses, keep
