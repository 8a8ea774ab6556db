"""TadGAN / HypAD networks on MI355X (reference: models/tadgan.py).

Same class names, constructor signatures, ``state_dict`` keys and output shapes as the reference; each
network's weights live in one flat device arena (hypad_amd/arena.py) consumed by the HIP kernels.
Training goes through the fused iteration functions of ``hypad_amd.train`` exactly where the reference calls
``loss.backward(); optim.step()``.  ``forward`` itself has two forms: one fused inference kernel when nothing can be
differentiated (``torch.no_grad()``, ``requires_grad_(False)``: the test loop, scoring), and -- when autograd is recording and a
parameter or the input requires a gradient, as in the reference's default state -- a chain of differentiable layer kernels
(``hypad_amd/autograd.py``), so that a caller's own ``loss.backward(); optimizer.step()`` works as with the reference.

The LSTMs run the reference's effective configuration: the window is the feature axis, sequence length is 1
and h0 = c0 = 0 (models/tadgan.py:24-25,59-60), so ``weight_hh`` is carried (and decayed by
RiemannianAdam) but never read by a forward.
"""
import itertools

import torch
from torch import nn

from .. import _C
from .. import autograd as hag
from ..arena import ArenaModule
from ..hyperspace.hyrnn_nets import MobiusLinear
from .init import linear_init, lstm_init

_tick = itertools.count(1)


def _drop(module, masks=None):
    d = _C.Dropout()
    d.train_mode = int(module.training)
    d.masks = None if masks is None else masks.data_ptr()
    d.seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    d.offset = next(_tick)
    return d


def _rows(x, width):
    x = x.reshape(-1, width)
    if not x.is_cuda:
        raise _C.HypadError("inputs must be on the GPU (hypad_amd has no CPU path)")
    return x.to(torch.float32).contiguous()


def _named(module):
    return {k: v for k, v in module.state_dict().items()}


class Encoder(ArenaModule):
    def __init__(self, signal_shape=100, latent_space_dim=20, hyperbolic=False):
        super().__init__()
        self.signal_shape, self.latent_space_dim = signal_shape, latent_space_dim
        # same construction (and RNG consumption) order as models/tadgan.py:15-21
        # (init.py: the draws of nn.LSTM(signal_shape, 50, 1, bidirectional=True) and nn.Linear(100, latent), without the modules)
        init = {f"lstm.{k}": v for k, v in lstm_init(signal_shape, 50, 1, True).items()}
        init.update({f"dense.{k}": v for k, v in linear_init(100, latent_space_dim).items()})
        self._init_arena(_C.NET_ENCODER, signal_shape, latent_space_dim, False, init)

    def forward(self, x):
        if hag.wants_graph(self, x):
            return hag.encoder_forward(self, _rows(x, self.signal_shape)).view(1, -1, self.latent_space_dim)
        x = _rows(x, self.signal_shape)
        out = torch.empty(x.shape[0], self.latent_space_dim, device=x.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_encoder_fwd(_C.ptr(self.arena()), _C.ptr(x), _C.ptr(out), x.shape[0], self.signal_shape,
                                          self.latent_space_dim, _C.stream()), "encoder_fwd")
        return out.view(1, -1, self.latent_space_dim)


class Decoder(ArenaModule):
    def __init__(self, signal_shape=100, latent_space_dim=20, hyperbolic=False):
        super().__init__()
        self.signal_shape, self.latent_space_dim, self.hyperbolic = signal_shape, latent_space_dim, hyperbolic
        init = {f"dense1.{k}": v for k, v in linear_init(latent_space_dim, 50).items()}        # models/tadgan.py:34
        init.update({f"lstm.{k}": v for k, v in lstm_init(50, 64, 2, True).items()})          # nn.LSTM(50, 64, num_layers=2, dropout=0.2, bidirectional=True)
        init.update({f"dense2.{k}": v for k, v in linear_init(128, signal_shape).items()})
        from ..arena import _Group
        self.dense1, self.lstm, self.dense2 = _Group(), _Group(), _Group()      # registration order = the reference's
        if hyperbolic:
            self.hyperbolic_linear = MobiusLinear(signal_shape, signal_shape, hyperbolic_input=False, hyperbolic_bias=True,
                                                  nonlin=None, fp64_hyper=False)                # models/tadgan.py:43-52
            init["hyperbolic_linear.weight"] = self.hyperbolic_linear.weight
            init["hyperbolic_linear.bias"] = self.hyperbolic_linear.bias
        self._init_arena(_C.NET_DECODER, signal_shape, latent_space_dim, hyperbolic, init)

    def forward(self, x, dropout_mask=None):
        if dropout_mask is None and hag.wants_graph(self, x):
            S = self.signal_shape
            out = hag.decoder_forward(self, _rows(x, self.latent_space_dim))
            return (out[0].view(1, -1, S), out[1].view(1, -1, S)) if self.hyperbolic else out.view(1, -1, S)
        z = _rows(x, self.latent_space_dim)
        rows, S = z.shape[0], self.signal_shape
        eucl = torch.empty(rows, S, device=z.device, dtype=torch.float32)
        hyper = torch.empty(rows, S, device=z.device, dtype=torch.float32) if self.hyperbolic else None
        d = _drop(self, dropout_mask)
        _C.check(_C.lib.hypad_decoder_fwd(_C.ptr(self.arena()), _C.ptr(z), _C.ptr(hyper), _C.ptr(eucl), rows, S,
                                          self.latent_space_dim, int(self.hyperbolic), d, _C.stream()), "decoder_fwd")
        if self.hyperbolic:
            return hyper.view(1, -1, S), eucl.view(1, -1, S)
        return eucl.view(1, -1, S)


class CriticX(ArenaModule):
    def __init__(self, signal_shape=10, latent_space_dim=20):
        super().__init__()
        self.signal_shape, self.latent_space_dim = signal_shape, latent_space_dim
        self.dropout = nn.Dropout(p=0.25)
        self.leakyrelu = nn.LeakyReLU(0.2)
        dims = [(signal_shape, latent_space_dim)] + [(latent_space_dim, latent_space_dim)] * 3 + [(latent_space_dim, 1)]
        init = {}
        for i, (a, b) in enumerate(dims, 1):                                                    # models/tadgan.py:77-89
            init.update({f"dense{i}.{k}": v for k, v in linear_init(a, b).items()})
        self._init_arena(_C.NET_CRITIC_X, signal_shape, latent_space_dim, False, init)

    def forward(self, x, dropout_masks=None):
        if dropout_masks is None and hag.wants_graph(self, x):
            return hag.critic_forward(self, _rows(x, self.signal_shape), 4, 0.25).view(1, -1, 1)
        x = _rows(x, self.signal_shape)
        out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_critic_x_fwd(_C.ptr(self.arena()), _C.ptr(x), _C.ptr(out), x.shape[0], self.signal_shape,
                                           self.latent_space_dim, _drop(self, dropout_masks), _C.stream()), "critic_x_fwd")
        return out.view(1, -1, 1)


class CriticZ(ArenaModule):
    def __init__(self, latent_space_dim=20):
        super().__init__()
        self.latent_space_dim = latent_space_dim
        init = {}
        for i, (a, b) in enumerate([(latent_space_dim, latent_space_dim)] * 2 + [(latent_space_dim, 1)], 1):
            init.update({f"dense{i}.{k}": v for k, v in linear_init(a, b).items()})            # models/tadgan.py:113-119
        self.dropout = nn.Dropout(p=0.2)
        self.leakyrelu = nn.LeakyReLU(0.2)
        self._init_arena(_C.NET_CRITIC_Z, latent_space_dim, latent_space_dim, False, init)

    def forward(self, x, dropout_masks=None):
        if dropout_masks is None and hag.wants_graph(self, x):
            return hag.critic_forward(self, _rows(x, self.latent_space_dim), 2, 0.2).view(*x.shape[:-1], 1)
        lead = x.shape[:-1]
        z = _rows(x, self.latent_space_dim)
        out = torch.empty(z.shape[0], device=z.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_critic_z_fwd(_C.ptr(self.arena()), _C.ptr(z), _C.ptr(out), z.shape[0], self.latent_space_dim,
                                           _drop(self, dropout_masks), _C.stream()), "critic_z_fwd")
        return out.view(*lead, 1)
