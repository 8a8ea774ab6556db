"""Round 6 diagnosis: bench_signals(WADI) itself, instrumented."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(100)


def run(S, warmup=2, steps=6):
    cfg = bench.Cfg("x", S=S, B=64, n_windows=20480, data="uniform")
    eng, x = bench.build_engine(1, 0, True, dev, cfg)
    step, losses = bench.make_step(eng, x, 1, gen, dev, graph=True, cfg=cfg)
    nb = cfg.nb
    for rnd in range(2):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        st = eng.status()
        l = losses.cpu().numpy()[0]
        bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
        info = None
        if len(bad):
            info = (int(bad[0]), len(bad), "gen" if bad[0] >= 10 * nb else ("cx" if bad[0] % 2 == 0 else "cz"), l[max(0, bad[0] - 1): bad[0] + 2].tolist())
        print("S", S, "round", rnd, "ms/epoch %.2f" % ms, "status", st, "counters", eng.counters.cpu().tolist(), "non-finite:", info, flush=True)
        if info:
            for net in ("enc", "dec", "cx", "cz"):
                sd = eng.state_dict(net, 0)
                print("   ", net, {k: bool(torch.isfinite(v).all()) for k, v in sd.items() if not bool(torch.isfinite(v).all())})
            break


for S in [int(a) for a in sys.argv[1:]]:
    run(S)
