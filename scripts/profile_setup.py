"""Where the set-up of the product loops goes: cProfile of train_signals_resident (32 models, 2 epochs, no files) and of
train.train over a DataLoader (configs[1], 3 epochs), sorted by cumulative time.   python scripts/profile_setup.py [n_signals]"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hypad_amd import train as ht  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.cuda.init(); torch.zeros(1, device="cuda")
data = [bench.synth_windows(bench.N_WINDOWS, bench.S, s) for s in range(n)]
P = lambda: SimpleNamespace(batch_size=64, signal_shape=100, latent_space_dim=20, lr=5e-4, hyperbolic=True, epochs=2, dataset="bench", signal="s", resume=False, resume_epoch=0)
os.chdir(tempfile.mkdtemp())
for rep in range(3):
    t0 = time.perf_counter()
    ht.train_signals_resident(data, P(), seed=1, log=None, save=False)
    torch.cuda.synchronize()
    print("train_signals_resident(%d signals, 2 epochs) plain call %d: %.1f ms" % (n, rep, 1e3 * (time.perf_counter() - t0)))
for rep in range(1):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    ht.train_signals_resident(data, P(), seed=1, log=None, save=False)
    torch.cuda.synchronize()
    pr.disable()
    print("train_signals_resident(%d signals, 2 epochs) call %d: %.1f ms" % (n, rep, 1e3 * (time.perf_counter() - t0)))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(s.getvalue().splitlines()[:60]))
