"""Per-epoch wall time inside one train_tadgan_resident call (log to log), with the reference's checkpoint cadence."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import bench
from hypad_amd import train as ht
from hypad_amd.models import tadgan
S, L, B = 100, 20, 64
P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, resume=False, resume_epoch=0)
torch.manual_seed(0)
mods = [m.cuda().train() for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, True), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
stamps = []
with tempfile.TemporaryDirectory() as d:
    ht.train_tadgan_resident(bench.synth_windows(1916, S, 0), *mods, n_epochs=44, params=P, path=d, seed=3, log=lambda s: stamps.append(time.perf_counter()))
    files = sorted(os.listdir(d))
w = np.diff(np.asarray(stamps)) * 1e3
print(" ".join("%d:%.2f" % (i + 1, v) for i, v in enumerate(w)))
print("median %.3f ms; mean from epoch 3 on %.3f ms; %d files" % (np.median(w), w[2:].mean(), len(files)))
