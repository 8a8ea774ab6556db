"""Shared helpers for the test-suite (fixtures -> oracle modules)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def sub_state(fx, prefix, wkey=None):
    """Extract one network's state_dict from a fixture ('enc.' keys, or 'w0.enc.' with wkey='w0')."""
    pre = (wkey + "." if wkey else "") + prefix + "."
    return {k[len(pre):]: torch.from_numpy(np.array(v)) for k, v in fx.items() if k.startswith(pre)}


def oracle_models(fx, S, hyperbolic=True, wkey=None, L=20):
    from oracle import tadgan
    enc = tadgan.Encoder(S, L)
    dec = tadgan.Decoder(S, L, hyperbolic)
    cx = tadgan.CriticX(S, L)
    cz = tadgan.CriticZ(L)
    enc.load_state_dict(sub_state(fx, "enc", wkey))
    dsd = sub_state(fx, "dec", wkey)
    if not hyperbolic:
        dsd = {k: v for k, v in dsd.items() if not k.startswith("hyperbolic_linear")}
    dec.load_state_dict(dsd)
    cx.load_state_dict(sub_state(fx, "cx", wkey))
    cz.load_state_dict(sub_state(fx, "cz", wkey))
    return enc, dec, cx, cz


def params_ns(B=64, S=100, hyperbolic=True, lr=5e-4):
    return SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=20, lr=lr, hyperbolic=hyperbolic)


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a


def maxdiff(a, b):
    a, b = np.asarray(_np(a), dtype=np.float64), np.asarray(_np(b), dtype=np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0
