"""A longer train_signals_resident run (32 signals, default 300 epochs, checkpoint cadence on): per-epoch wall by quarter, files written, finite histories."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from types import SimpleNamespace
from hypad_amd import train as ht
T = {"layout": [0.0, 0], "torch.save": [0.0, 0], "submit": [0.0, 0], "snapshot": [0.0, 0]}
def _timed(key, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key][0] += time.perf_counter() - t0; T[key][1] += 1; return r
    return g
ht._SavedLayout.write = _timed("layout", ht._SavedLayout.write)
torch.save = _timed("torch.save", torch.save)
ht._CheckpointWriter.submit = _timed("submit", ht._CheckpointWriter.submit)
import queue as _q
T["put"] = [0.0, 0]; T["get"] = [0.0, 0]
_q.Queue.put = _timed("put", _q.Queue.put)
_q.Queue.get = _timed("get", _q.Queue.get)
ht._CheckpointWriter.snapshot = _timed("snapshot", ht._CheckpointWriter.snapshot)
n_sig, epochs = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 300
if len(sys.argv) > 2:
    sys.setswitchinterval(float(sys.argv[2]))
datasets = [bench.synth_windows(1916, 100, s) for s in range(n_sig)]
with tempfile.TemporaryDirectory() as d:
    os.chdir(d)
    P = SimpleNamespace(batch_size=64, signal_shape=100, latent_space_dim=20, lr=5e-4, hyperbolic=True, epochs=epochs, dataset="soak", signal="s", resume=False, resume_epoch=0)
    stamps, intervals = [], []
    res = ht.train_signals_resident(datasets, P, seed=1, log=lambda s: (stamps.append(time.perf_counter()), intervals.append(sys.getswitchinterval())), save=True)
    print("switch interval seen by the log calls:", sorted(set(intervals)))
    files = sum(len(f) for _, _, f in os.walk(d))
    os.chdir("/tmp")
w = np.diff(np.asarray(stamps)) * 1e3
q = len(w) // 4
ok = all(np.isfinite(r["history"]["dec"]).all() and len(r["history"]["dec"]) == epochs for r in res.values())
print("%d signals x %d epochs: ms per epoch by quarter %s (longest %.1f); %d files; histories finite and complete: %s; device memory reserved %.0f MB" % (
    n_sig, epochs, " ".join("%.3f" % w[i * q:(i + 1) * q].mean() for i in range(4)), w.max(), files, ok, torch.cuda.memory_reserved() / 2**20))
print({k: "%.1f ms in %d calls" % (1e3 * v[0], v[1]) for k, v in T.items()})
