"""Round 6 diagnosis: the first non-finite loss row of the WADI-shaped run whose shuffles come from an advanced generator; state saved for a CPU replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(100)
for _ in range(16):                                   # what the S = 51 section drew before
    torch.rand(6, 20480, device=dev, generator=gen)
S = 123
cfg = bench.Cfg("x", S=S, B=64, n_windows=20480, data="uniform")
eng, x = bench.build_engine(1, 0, True, dev, cfg)
nb = cfg.nb
losses = torch.empty(1, 11 * nb, 4, device=dev)
perm_buf = torch.empty(6, nb * 64, dtype=torch.int32, device=dev)
mode = os.environ.get("MODE", "graph")
for ep in range(16):
    before = {net: {k: v.clone() for k, v in eng.state_dict(net, 0).items()} for net in ("enc", "dec", "cx", "cz")}
    perm = torch.rand(6, 20480, device=dev, generator=gen).argsort(dim=1)[:, : nb * 64]
    perm_buf.copy_(perm)
    if mode == "graph":
        eng.train_epoch_graph(x, perm_buf, nb, 5, train_mode=True, losses=losses, shuffle_windows=0)
    else:
        eng.train_epoch(x, perm_buf, nb, 5, train_mode=True, losses=losses, flags=int(os.environ.get("FLAGS", "0")))
    torch.cuda.synchronize()
    l = losses.cpu().numpy()[0]
    bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
    print("epoch", ep, "status", eng.status(), "bad rows", len(bad), "first", (int(bad[0]) if len(bad) else None), "cx mean", float(np.nanmean(l[:10 * nb:2, 0])), flush=True)
    if len(bad):
        b = int(bad[0])
        print("  kind", "gen" if b >= 10 * nb else ("cx" if b % 2 == 0 else "cz"), "iteration", b // 2 if b < 10 * nb else b - 10 * nb, "rows", l[max(0, b - 2): b + 3].tolist())
        os.makedirs("gpurun_out", exist_ok=True)
        torch.save({"before": {n: {k: v.cpu() for k, v in sd.items()} for n, sd in before.items()}, "perm": perm_buf.cpu(), "losses": torch.from_numpy(l), "bad": b,
                    "x_seed": 0, "epoch": ep}, "gpurun_out/wadi_nan_state.pt")
        break
