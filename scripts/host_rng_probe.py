"""Host-side cost of the reference's per-iteration random draws (train.py:24,64,118,149,205) on this machine."""
import time, os, threading, numpy as np, torch
def t(f, n=300):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
print("np normal (1,64,20) us", t(lambda: np.random.normal(size=(1, 64, 20))))
print("np normal (319,64,20) us", t(lambda: np.random.normal(size=(319, 64, 20)), 20))
print("torch rand (1,64,100) us", t(lambda: torch.rand((1, 64, 100))))
print("torch rand (29*7680,) us", t(lambda: torch.rand((29 * 7680,)), 50))
buf = torch.empty(145 * 7680).pin_memory() if torch.cuda.is_available() else torch.empty(145 * 7680)
print("torch rand out= pinned (29*7680) us", t(lambda: torch.rand((29 * 7680,), out=buf[:29 * 7680]), 50))
def both():
    th = threading.Thread(target=lambda: np.random.normal(size=(319, 64, 20)))
    th.start()
    for _ in range(5): torch.rand((29 * 7680,))
    th.join()
print("np (319 iters) || torch (5 passes) us", t(both, 20))
from torch.utils.data import DataLoader
X = np.random.rand(1916, 100, 1)
class DS:
    def __len__(self): return len(X)
    def __getitem__(self, i): return torch.from_numpy(X[i])
dl = DataLoader(DS(), batch_size=64, shuffle=True, drop_last=True, num_workers=0)
def it():
    for s in dl: pass
print("DataLoader pass (29 batches, workers=0) us", t(it, 20))
import sys; sys.path.insert(0, "."); from hypad_amd import host_rng
zx = np.zeros((145, 1280), np.float32); zz = np.zeros_like(zx); zg = np.zeros((29, 1280), np.float32)
def native():
    host_rng.global_normal_into([zx, zz], 1280, 145); host_rng.global_normal_into([zg], 1280, 29)
best = min(t(native, 10) for _ in range(5))
print("native MT19937 normal, one epoch's 408 320 draws us (best of 5 x 10)", best)
print("np normal (319,64,20) us (best of 5 x 10)", min(t(lambda: np.random.normal(size=(319, 64, 20)), 10) for _ in range(5)))
