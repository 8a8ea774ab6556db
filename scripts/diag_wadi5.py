"""Round 6 diagnosis: which ingredient makes queued WADI-shaped epochs go non-finite.  usage: diag_wadi5.py S N mode(graph|eager|periter) train(1|0) queued_epochs rounds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from hypad_amd import _C

S, N, mode, train, queued, rounds = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1", int(sys.argv[5]), int(sys.argv[6])
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(100)
cfg = bench.Cfg("x", S=S, B=64, n_windows=N, data="uniform")
eng, x = bench.build_engine(1, 0, True, dev, cfg)
nb = cfg.nb
losses = torch.empty(1, 11 * nb, 4, device=dev)
perm_buf = torch.empty(6, nb * 64, dtype=torch.int32, device=dev)
flags = _C.EPOCH_PER_ITERATION if mode == "periter" else 0
res = []
for r in range(rounds):
    for ep in range(queued):
        perm = torch.rand(6, N, device=dev, generator=gen).argsort(dim=1)[:, : nb * 64]
        perm_buf.copy_(perm)
        if mode == "graph":
            eng.train_epoch_graph(x, perm_buf, nb, 5, train_mode=train, losses=losses, shuffle_windows=0)
        else:
            eng.train_epoch(x, perm_buf, nb, 5, train_mode=train, losses=losses, flags=flags)
    torch.cuda.synchronize()
    l = losses.cpu().numpy()[0]
    bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
    res.append((len(bad), int(bad[0]) if len(bad) else None))
    if len(bad):
        break
print(" ".join(sys.argv[1:]), "status", eng.status(), "census", int(eng.counters[5]), "per round (bad rows, first):", res, flush=True)
