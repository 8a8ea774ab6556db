"""Round 6 diagnosis: where train.train's set-up time goes (bench.py call_level: 15 ms in round 5, 68 ms alone / 870 ms inside the full run now)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(100)
r = bench.bench_call_level(dev, epochs=40)
print("call_level alone", {k: round(r[k], 1) for k in ("first_call_ms", "call_ms", "setup_ms", "ms_per_epoch", "after_last_epoch_ms")}, flush=True)
pr = cProfile.Profile()
pr.enable()
r = bench.bench_call_level(dev, epochs=3)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000], flush=True)
for what in sys.argv[1:]:
    if what == "wadi":
        cfg = bench.Cfg("w", S=123, B=64, n_windows=20480, data="uniform")
        print(bench.RankGuard().run(bench.bench_signals, 1, 0, dev, gen, warmup=2, steps=6, cfg=cfg, eager=False).get("error", "wadi ok"))
    elif what == "swat":
        cfg = bench.Cfg("w", S=51, B=64, n_windows=20480, data="uniform")
        print(bench.RankGuard().run(bench.bench_signals, 1, 0, dev, gen, warmup=2, steps=6, cfg=cfg, eager=False).get("error", "swat ok"))
    elif what == "s32":
        print(bench.RankGuard().run(bench.bench_signals, 32, 0, dev, gen, warmup=2, steps=8, eager=False).get("error", "s32 ok"))
    elif what == "s32p":
        print(str(bench.RankGuard().run(bench.bench_signals_product, 32, dev))[:200])
    r = bench.bench_call_level(dev, epochs=40)
    print("call_level after", what, {k: round(r[k], 1) for k in ("first_call_ms", "call_ms", "setup_ms", "ms_per_epoch", "after_last_epoch_ms")}, flush=True)
