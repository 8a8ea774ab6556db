"""hypad_amd.train.train_signals_resident (the product loop over many signals: per-signal histories, checkpoint cadence) against the bare
engine replay bench.py's `signals32` section times: ms per epoch of all signals."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from types import SimpleNamespace
from hypad_amd import train as ht
n_sig = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S, B = 100, 64
datasets = [bench.synth_windows(1916, S, s) for s in range(n_sig)]
for epochs, save in ((24, False), (24, True)):
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)                      # (model_path is relative to the working directory, as in the reference)
        P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=20, lr=5e-4, hyperbolic=True, epochs=epochs, dataset="bench", signal="s", model_path=d,
                            resume=False, resume_epoch=0, new_features=False, id=0)
        stamps = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ht.train_signals_resident(datasets, P, seed=1, log=lambda s: stamps.append(time.perf_counter()), save=save)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        os.chdir("/tmp")
    w = np.diff(np.asarray(stamps)) * 1e3
    print("%d signals, %d epochs, save=%s: %.1f ms in all; set-up + first epoch %.1f ms; after the last epoch's log %.1f ms" % (n_sig, epochs, save, 1e3 * dt, 1e3 * (stamps[0] - t0), 1e3 * (t0 + dt - stamps[-1])))
    print("   per epoch (log to log): median %.3f ms = %.2f M windows/s; mean from epoch 3 on %.3f ms; longest %.1f ms" % (np.median(w), n_sig * 29 * B / np.median(w) / 1e3, w[2:].mean(), w.max()), flush=True)
