"""Round 6 diagnosis: non-finite losses of the WADI-shaped epoch (window 123, batch 64, 20 480 windows U(-1, 1)) after ~10 epochs in bench.py's
multivariate_wadi section: kernel fault or training dynamics?  Neighbouring widths (run-time-shape kernels) and the per-iteration form beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from hypad_amd import _C

dev = torch.device("cuda", 0)
EPOCHS = int(os.environ.get("EPOCHS", "24"))
GRAPH = os.environ.get("GRAPH", "1") == "1"
for S, Bc, train_mode, flags in ((123, 64, True, 0), (123, 64, False, 0), (124, 64, True, 0), (51, 64, True, 0), (100, 64, True, 0)):
    gen = torch.Generator(device=dev).manual_seed(100)
    cfg = bench.Cfg("x", S=S, B=Bc, n_windows=20480, data="uniform")
    eng, x = bench.build_engine(1, 0, True, dev, cfg)
    eng.epoch_flags = flags
    nb = cfg.nb
    losses = torch.empty(1, 11 * nb, 4, device=dev)
    perm_buf = torch.empty(6, nb * Bc, dtype=torch.int32, device=dev)
    first, traj = None, []
    for ep in range(EPOCHS):
        perm = torch.rand(6, 20480, device=dev, generator=gen).argsort(dim=1)[:, : nb * Bc]
        perm_buf.copy_(perm)
        if GRAPH:
            eng.train_epoch_graph(x, perm_buf, nb, 5, train_mode=train_mode, losses=losses, shuffle_windows=0)
        else:
            eng.train_epoch(x, perm_buf, nb, 5, train_mode=train_mode, losses=losses, flags=flags)
        torch.cuda.synchronize()
        l = losses.cpu().numpy()[0]
        g = l[10 * nb:]
        traj.append((round(float(np.nanmean(l[:10 * nb:2, 0])), 3), round(float(np.nanmean(g[:, 0])), 3), round(float(np.nanmax(np.abs(g[:, 1]))), 4)))
        bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
        if len(bad):
            first = (ep, int(bad[0]), "gen" if bad[0] >= 10 * nb else ("cx" if bad[0] % 2 == 0 else "cz"), l[max(0, bad[0] - 2): bad[0] + 2].tolist())
            break
    norms = {k: float(v.abs().max()) for k, v in eng.state_dict("dec", 0).items() if "hyperbolic" in k}
    print("S", S, "B", Bc, "train" if train_mode else "eval", "flags", flags, "status", eng.status(), "first non-finite:", first, "\n   (cx mean, gen mean, max aux) per epoch:", traj[-8:],
          "head |w|max", norms, flush=True)
    del eng, x
