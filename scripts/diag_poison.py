"""Round 6 diagnosis: does an epoch read workspace it never wrote?  Fill the caching allocator's free blocks with a NaN (or huge-value) pattern, then build the
engine in them and train."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

dev = torch.device("cuda", 0)
EPOCHS = int(os.environ.get("EPOCHS", "16"))
pattern = os.environ.get("PATTERN", "nan")
for S, Bc, N in ((123, 64, 20480), (51, 64, 20480), (100, 64, 1916), (150, 256, 20480), (123, 64, 1916), (123, 64, 8192)):
    for poison in (False, True):
        torch.cuda.empty_cache()
        if poison:
            blocks = [torch.empty(n, device=dev) for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
            for b in blocks:
                if pattern == "nan":
                    b.fill_(float("nan"))
                else:
                    b.view(torch.int32).fill_(0x7f7fffff if pattern == "max" else 0x5f5f5f5f)
            torch.cuda.synchronize()
            del blocks                       # back to the caching allocator, contents kept
        gen = torch.Generator(device=dev).manual_seed(100)
        cfg = bench.Cfg("x", S=S, B=Bc, n_windows=N, data="uniform")
        eng, x = bench.build_engine(1, 0, True, dev, cfg)
        step, losses = bench.make_step(eng, x, 1, gen, dev, graph=True, cfg=cfg)
        first = None
        for ep in range(EPOCHS):
            step()
            torch.cuda.synchronize()
            l = losses.cpu().numpy()[0]
            bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
            if len(bad):
                nb = cfg.nb
                first = (ep, int(bad[0]), "gen" if bad[0] >= 10 * nb else ("cx" if bad[0] % 2 == 0 else "cz"), l[max(0, bad[0] - 1): bad[0] + 2].tolist())
                break
        print("S", S, "B", Bc, "N", N, "poisoned" if poison else "clean", "status", eng.status(), "first non-finite:", first, flush=True)
        del eng, x, step, losses
