"""Opt-in timing check (not part of `pytest -m gpu`): hypad_critic_x_iteration as a one-iteration phase of the hoisted critic form
against the stand-alone launches -- measured 44.5-45.0 against 63.3-63.7 us of GPU time per call.  Exits non-zero if the phase form is
not at least 5 % faster.      python scripts/check_phase_form_speed.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hypad_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda", 0)
out = {}
for on in (True, False):
    Engine.iteration_phase = on
    eng, x = bench.build_engine(1, 0, True, dev)
    xb = x[:, :64].contiguous()
    z, ax = torch.randn(1, 64, 20, device=dev), torch.rand(1, 64, 100, device=dev)
    for _ in range(3):
        eng.critic_x_iteration(xb, None, z, ax, train_mode=False)
    best = float("inf")
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(20_000_000)
        a.record()
        for _ in range(10):
            eng.critic_x_iteration(xb, None, z, ax, train_mode=False)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10 * 1e3)
    out[on] = best
print("critic_x GPU us per call: phase form %.1f, stand-alone launches %.1f" % (out[True], out[False]))
sys.exit(0 if out[True] < 0.95 * out[False] else 1)
