"""Where the host time of anomaly_detection.test_tadgan goes (cProfile, 125 000 windows): tensor dataset vs SignalDataset."""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import numpy as np, pandas as pd, torch
from types import SimpleNamespace
from torch.utils.data import DataLoader
from hypad_amd import anomaly_detection as had
from hypad_amd.models.tadgan import Encoder, Decoder, CriticX
from hypad_amd.utils.dataloader import SignalDataset
S, L, B = 100, 20, 64
P = SimpleNamespace(batch_size=B, signal_shape=S, hyperbolic=True)
torch.manual_seed(0)
enc, dec, cx = Encoder(S, L).cuda(), Decoder(S, L, True).cuda(), CriticX(S, L).cuda()
tt = np.arange(125_000 + S)
sds = SignalDataset(pd.DataFrame({"timestamp": 1_400_000_000 + 600 * tt, "value": np.sin(tt / 50.0)}), interval=600, windows_size=S, test=True)
big = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (125_000, S, 1)))
for name, ds in (("tensor", big), ("signal", sds)):
    loader = DataLoader(ds, batch_size=B, shuffle=False)
    for _ in range(2):
        had.test_tadgan(loader, enc, dec, cx, path="", signal_shape=S, params=P)
    t0 = time.perf_counter(); had.test_tadgan(loader, enc, dec, cx, path="", signal_shape=S, params=P); print(name, "%.2f ms" % (1e3 * (time.perf_counter() - t0)))
    pr = cProfile.Profile(); pr.enable()
    had.test_tadgan(loader, enc, dec, cx, path="", signal_shape=S, params=P)
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = had.score_batches(loader, enc, dec, cx, S)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        h = had._to_host({"recons": res["recons"], "critic": res["critic"], "hyper_real": res["hyper_real"]})
        t3 = time.perf_counter()
        print(name, "score_batches host %.2f ms, device drain %.2f ms, to_host %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
