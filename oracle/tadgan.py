"""TadGAN / HypAD networks on CPU (oracle; test infrastructure).

Restates ``models/tadgan.py`` and ``hyperspace/hyrnn_nets.py:154-200`` with the
same ``state_dict`` keys so weights move freely between the
reference, this oracle and the HIP path.  ``nn.LSTM``/``nn.Linear`` are kept on
purpose: timed on host cores this is the reference's CPU execution profile.
"""
import math

import torch
from torch import nn

from . import gmath


class BallParameter(nn.Parameter):
    """Marks a tensor as Poincare-ball valued (the reference's geoopt.ManifoldParameter,
    hyperspace/hyrnn_nets.py:168-169)."""

    def __new__(cls, data, requires_grad=True):
        inst = nn.Parameter.__new__(cls, data, requires_grad)
        inst.manifold = "poincare_ball_c1"
        return inst


class MobiusLinear(nn.Linear):
    """hyperspace/hyrnn_nets.py:154-200 in the configuration of models/tadgan.py:43-52."""

    def __init__(self, in_features, out_features):
        super().__init__(in_features, out_features)
        with torch.no_grad():
            ball_bias = gmath.expmap0(torch.randn(out_features) / 400)      # hyrnn_nets.py:173
            std = 1.0 / math.sqrt(2 * out_features * in_features) / 100     # hyrnn_nets.py:176-178
            self.weight.normal_(std=std)
        self.bias = BallParameter(ball_bias)

    def forward(self, inp):
        return gmath.mobius_linear(inp.float(), self.weight, self.bias)     # hyrnn_nets.py:186-200


class Encoder(nn.Module):
    """models/tadgan.py:10-27"""

    def __init__(self, signal_shape=100, latent_space_dim=20, hyperbolic=False):
        super().__init__()
        self.signal_shape, self.latent_space_dim = signal_shape, latent_space_dim
        self.lstm = nn.LSTM(input_size=signal_shape, hidden_size=50, num_layers=1, bidirectional=True)
        self.dense = nn.Linear(100, latent_space_dim)

    def forward(self, x):
        h, _ = self.lstm(x.view(1, -1, self.signal_shape).float())
        return self.dense(h)


class Decoder(nn.Module):
    """models/tadgan.py:30-67"""

    def __init__(self, signal_shape=100, latent_space_dim=20, hyperbolic=False):
        super().__init__()
        self.signal_shape, self.latent_space_dim, self.hyperbolic = signal_shape, latent_space_dim, hyperbolic
        self.dense1 = nn.Linear(latent_space_dim, 50)
        self.lstm = nn.LSTM(input_size=50, hidden_size=64, num_layers=2, dropout=0.2, bidirectional=True)
        self.dense2 = nn.Linear(128, signal_shape)
        if hyperbolic:
            self.hyperbolic_linear = MobiusLinear(signal_shape, signal_shape)

    def forward(self, z):
        h, _ = self.lstm(self.dense1(z))
        e = torch.tanh(self.dense2(h))
        if self.hyperbolic:
            return self.hyperbolic_linear(e.view(-1, self.signal_shape)).view(1, -1, self.signal_shape), e
        return e


class _Critic(nn.Module):
    def _mlp(self, x):
        for name in self._hidden:
            x = self.dropout(nn.functional.leaky_relu(getattr(self, name)(x), 0.2))
        return getattr(self, self._last)(x)


class CriticX(_Critic):
    """models/tadgan.py:70-106"""
    _hidden, _last = ("dense1", "dense2", "dense3", "dense4"), "dense5"

    def __init__(self, signal_shape=10, latent_space_dim=20):
        super().__init__()
        self.signal_shape, self.latent_space_dim = signal_shape, latent_space_dim
        self.dropout = nn.Dropout(0.25)
        self.dense1 = nn.Linear(signal_shape, latent_space_dim)
        self.dense2 = nn.Linear(latent_space_dim, latent_space_dim)
        self.dense3 = nn.Linear(latent_space_dim, latent_space_dim)
        self.dense4 = nn.Linear(latent_space_dim, latent_space_dim)
        self.dense5 = nn.Linear(latent_space_dim, 1)

    def forward(self, x):
        return self._mlp(x.view(1, -1, self.signal_shape).float())


class CriticZ(_Critic):
    """models/tadgan.py:109-132"""
    _hidden, _last = ("dense1", "dense2"), "dense3"

    def __init__(self, latent_space_dim=20):
        super().__init__()
        self.latent_space_dim = latent_space_dim
        self.dense1 = nn.Linear(latent_space_dim, latent_space_dim)
        self.dense2 = nn.Linear(latent_space_dim, latent_space_dim)
        self.dense3 = nn.Linear(latent_space_dim, 1)
        self.dropout = nn.Dropout(0.2)

    def forward(self, x):
        return self._mlp(x)


def build_models(signal_shape=100, latent=20, hyperbolic=True, seed=0):
    """Construction order of train.py:415-426 under one manual seed."""
    torch.manual_seed(seed)
    enc = Encoder(signal_shape, latent).train()
    dec = Decoder(signal_shape, latent, hyperbolic).train()
    cx = CriticX(signal_shape, latent).train()
    cz = CriticZ(latent).train()
    return enc, dec, cx, cz
