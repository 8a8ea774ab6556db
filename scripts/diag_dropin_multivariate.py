"""The drop-in train_tadgan at configs[3]'s shape (window 150, batch 256, 20 480 windows): per-epoch wall inside one call, and the
producer's pieces."""
import sys, os, io, contextlib, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from hypad_amd import train as ht, epoch_feed
from hypad_amd.models import tadgan
S, L, B, N = 150, 20, 256, 20480
if "--switch" in sys.argv:
    sys.setswitchinterval(float(sys.argv[sys.argv.index("--switch") + 1]))
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for n in ("prepare", "_alphas", "_draw_z", "_pass_indices"):
    wrap(epoch_feed.EpochFeed, n)
data = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (N, S, 1)))
loader = torch.utils.data.DataLoader(data, batch_size=B, drop_last=True, shuffle=True)
P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, resume=False, resume_epoch=0)
torch.manual_seed(0); np.random.seed(0)
mods = [m.cuda().train() for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, True), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
    hist = ht.train_tadgan(loader, *mods, n_epochs=9, params=P, path=d)
w = np.diff(np.asarray(hist.wall)) * 1e3
print("epoch wall ms:", " ".join("%.1f" % v for v in w), " median %.2f ms = %.2f M windows/s" % (np.median(w), (N // B) * B / np.median(w) / 1e3))
for k, v in T.items():
    print("   %-14s n=%3d  median %.2f ms  per epoch %.1f ms" % (k, len(v), 1e3 * np.median(v), 1e3 * sum(v) / 9))
