"""Where an epoch of the drop-in train_tadgan spends its host time: prepare / upload + replay / finish, per epoch."""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import bench
from hypad_amd import train as ht, epoch_feed
from hypad_amd.models import tadgan
S, L, B = 100, 20, 64
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for n in ("get", "prepare", "upload", "_alphas", "_pass_samples", "_pass_indices", "_draw_z"):
    wrap(epoch_feed.EpochFeed, n)
from hypad_amd import engine
wrap(engine.Engine, "train_epoch_graph")
wrap(torch.cuda.Event, "synchronize")
data = torch.from_numpy(bench.synth_windows(1916, S, 0)[: 29 * B, :, None])
host_list = [data[b * B:(b + 1) * B] for b in range(29)]
for nthreads in (None, 1):
    if nthreads: torch.set_num_threads(nthreads)
    for loader, name in ((host_list, "list"), (torch.utils.data.DataLoader(bench._synthetic_signal_dataset(), batch_size=B, drop_last=True, shuffle=True), "dataloader")):
        T.clear()
        P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, resume=False, resume_epoch=0)
        torch.manual_seed(0); np.random.seed(0)
        mods = [m.cuda().train() for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, True), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            ht.train_tadgan(loader, *mods, n_epochs=24, params=P, path="/tmp")
        torch.cuda.synchronize()
        print(name, "threads", torch.get_num_threads(), "total ms", 1e3 * (time.perf_counter() - t0))
        for k, v in T.items():
            print("   %-20s n=%4d  median %.3f ms  last %.3f ms  sum %.1f ms" % (k, len(v), 1e3 * np.median(v), 1e3 * v[-1], 1e3 * sum(v)))
