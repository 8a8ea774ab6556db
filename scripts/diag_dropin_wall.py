"""Per-epoch wall time inside one drop-in train_tadgan call (hist.wall differences): which epochs are long?"""
import sys, os, io, contextlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import bench
from hypad_amd import train as ht
from hypad_amd.models import tadgan
S, L, B = 100, 20, 64
loader = torch.utils.data.DataLoader(bench._synthetic_signal_dataset(), batch_size=B, drop_last=True, shuffle=True)
P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, resume=False, resume_epoch=0)
torch.manual_seed(0); np.random.seed(0)
mods = [m.cuda().train() for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, True), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
    hist = ht.train_tadgan(loader, *mods, n_epochs=44, params=P, path=d)
w = np.diff(np.asarray(hist.wall)) * 1e3
print("epoch wall ms (epoch k+1's losses on the host minus epoch k's):")
print(" ".join("%d:%.2f" % (i + 1, v) for i, v in enumerate(w)))
plain = [v for i, v in enumerate(w) if (i + 1) % 10 not in (9, 0) and i >= 2]
print("median of epochs away from a checkpoint %.3f ms; mean of all from epoch 3 on %.3f ms" % (np.median(plain), np.mean(w[2:])))
