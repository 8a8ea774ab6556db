"""train_signals_resident over signals of many different lengths (few models per batch count: small groups): ms per epoch of all signals."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from types import SimpleNamespace
from hypad_amd import train as ht
B = 64
counts = [B * nb + 7 * (i % 5) for i, nb in enumerate([6, 6, 6, 9, 9, 9, 9, 12, 12, 12, 15, 15, 15, 15, 18, 18, 18, 21, 21, 21, 21, 24, 24, 24, 27, 27, 27, 27, 29, 29, 29, 29])]
datasets = [bench.synth_windows(n, 100, s) for s, n in enumerate(counts)]
plan, _ = ht.plan_signal_groups(counts, B)
print("groups:", [len(m) for _, m in plan])
for lanes in (1, None):
  with tempfile.TemporaryDirectory() as d:
    os.chdir(d)
    P = SimpleNamespace(lanes=lanes, batch_size=B, signal_shape=100, latent_space_dim=20, lr=5e-4, hyperbolic=True, epochs=40, dataset="ragged", signal="s", resume=False, resume_epoch=0)
    stamps = []
    res = ht.train_signals_resident(datasets, P, seed=1, log=lambda s: stamps.append(time.perf_counter()), save=False)
    os.chdir("/tmp")
  w = np.diff(np.asarray(stamps)) * 1e3
  ok = all(np.isfinite(r["history"]["dec"]).all() for r in res.values())
  hists = {n: r["history"] for n, r in res.items()}
  print("lanes", lanes or "auto", "| same histories as one lane:", hists == first if lanes is None else None)
  first = hists
  print("%d signals in %d groups: %.3f ms per epoch of all (median; mean from epoch 3 on %.3f) = %.2f M windows/s; finite %s" % (
    len(counts), len(plan), np.median(w), w[2:].mean(), sum(n // B * B for n in counts) / np.median(w) / 1e3, ok))
