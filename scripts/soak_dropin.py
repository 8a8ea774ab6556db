"""A longer drop-in run (default 1 500 epochs, the reference trains 2 000): per-epoch wall stays flat, host RSS and device memory do not grow."""
import sys, os, io, contextlib, tempfile, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import bench
from hypad_amd import train as ht
from hypad_amd.models import tadgan
n_epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
S, L, B = 100, 20, 64
loader = torch.utils.data.DataLoader(bench._synthetic_signal_dataset(), batch_size=B, drop_last=True, shuffle=True)
P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, resume=False, resume_epoch=0)
torch.manual_seed(0); np.random.seed(0)
mods = [m.cuda().train() for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, True), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
    hist = ht.train_tadgan(loader, *mods, n_epochs=n_epochs, params=P, path=d)
    files = len(os.listdir(d))
w = np.diff(np.asarray(hist.wall)) * 1e3
q = len(w) // 4
print("%d epochs, %d checkpoint files; ms per epoch by quarter: %s; longest %.1f ms" % (n_epochs, files, " ".join("%.3f" % w[i * q:(i + 1) * q].mean() for i in range(4)), w.max()))
print("final losses cx %.4f cz %.4f dec %.4f hyper %.5f (finite: %s)" % (hist.cx[-1], hist.cz[-1], hist.dec[-1], hist.hyper[-1], bool(np.isfinite(hist.dec).all())))
print("host max RSS %.0f -> %.0f MB; device memory allocated %.1f MB, reserved %.1f MB" % (rss0 / 1024, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024,
                                                                                      torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20))
