"""MobiusLinear / mobius_linear on the GPU (reference: hyperspace/hyrnn_nets.py:13-35, :154-200).

Only the configuration the hot path uses is implemented: Euclidean input, ball-valued bias, no
non-linearity, k = -1, fp32 (models/tadgan.py:43-52).  Anything else raises.
"""
import math

import torch
from torch import nn

from .. import _C


class PoincareBall:
    """Marker for ball-valued parameters (the reference's geoopt.PoincareBall(c=1))."""

    def __init__(self, c=1.0):
        self.c = torch.tensor(float(c))
        self.k = -self.c


class ManifoldParameter(nn.Parameter):
    """The reference's geoopt.ManifoldParameter: an nn.Parameter tagged with its manifold."""

    def __new__(cls, data=None, manifold=None, requires_grad=True):
        inst = nn.Parameter.__new__(cls, data, requires_grad)
        inst.manifold = manifold
        return inst

    def __reduce_ex__(self, proto):
        return _rebuild_manifold_parameter, (self.data, self.manifold, self.requires_grad)


def _rebuild_manifold_parameter(data, manifold, requires_grad):
    return ManifoldParameter(data, manifold=manifold, requires_grad=requires_grad)


class _MobiusLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _C.require_cuda(x.to(torch.float32).contiguous(), "input")
        w = _C.require_cuda(weight.contiguous(), "weight")
        b = _C.require_cuda(bias.contiguous(), "bias")
        x2 = x.reshape(-1, x.shape[-1])
        rows, k, n = x2.shape[0], x2.shape[1], w.shape[0]
        out = torch.empty(rows, n, device=x.device, dtype=torch.float32)
        u = torch.empty(rows, n, device=x.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_mobius_linear_fwd(_C.ptr(x2), _C.ptr(w), _C.ptr(b), _C.ptr(out), _C.ptr(u), rows, k, n, _C.stream()),
                 "mobius_linear_fwd")
        ctx.save_for_backward(x2, w, b, u)
        ctx.xshape = x.shape
        return out.view(*x.shape[:-1], n)

    @staticmethod
    @_C.first_order_only
    def backward(ctx, go):
        x2, w, b, u = ctx.saved_tensors
        rows, k, n = x2.shape[0], x2.shape[1], w.shape[0]
        go2 = go.to(torch.float32).contiguous().reshape(rows, n)
        gx = torch.empty_like(x2)
        gw = torch.empty_like(w)
        gb = torch.empty_like(b)
        nbytes = _C.lib.hypad_mobius_linear_workspace_bytes(rows, n)
        ws = torch.empty(max(nbytes // 4, 1), device=x2.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_mobius_linear_bwd(_C.ptr(x2), _C.ptr(w), _C.ptr(b), _C.ptr(u), _C.ptr(go2), _C.ptr(gx), _C.ptr(gw),
                                                _C.ptr(gb), _C.ptr(ws), nbytes, rows, k, n, _C.stream()), "mobius_linear_bwd")
        return gx.view(ctx.xshape), gw, gb


def mobius_linear(input, weight, bias=None, hyperbolic_input=True, hyperbolic_bias=True, nonlin=None, k=-1.0):
    if hyperbolic_input or not hyperbolic_bias or nonlin is not None or bias is None or abs(float(k) + 1.0) > 1e-12:
        raise NotImplementedError("mobius_linear: only hyperbolic_input=False, hyperbolic_bias=True, nonlin=None, k=-1 "
                                  "(the configuration of models/tadgan.py:43-52) runs on the HIP path")
    return _MobiusLinearFn.apply(input, weight, bias)


def _expmap0_host(u):
    """Initialisation-time expmap0 on the host (hyrnn_nets.py:173); the run-time op is gmath.expmap0."""
    n = u.norm(dim=-1, keepdim=True).clamp_min(1e-15)
    return torch.tanh(n.clamp(-15, 15)) * (u / n)


class MobiusLinear(nn.Linear):
    def __init__(self, *args, hyperbolic_input=True, hyperbolic_bias=True, nonlin=None, k=-1.0, fp64_hyper=True, **kwargs):
        super().__init__(*args, **kwargs)
        if fp64_hyper:
            raise NotImplementedError("fp64_hyper=True: the reference's hot path uses fp32 (models/tadgan.py:51)")
        if self.bias is not None and hyperbolic_bias:
            self.ball = PoincareBall(c=abs(float(k)))
            with torch.no_grad():
                ball_bias = _expmap0_host(self.bias.detach().clone().normal_() / 400)        # hyrnn_nets.py:173
            self.bias = ManifoldParameter(ball_bias, manifold=self.ball)
        with torch.no_grad():
            std = 1 / math.sqrt(2 * self.weight.shape[0] * self.weight.shape[1]) / 100       # hyrnn_nets.py:176-178
            self.weight.normal_(std=std)
        self.hyperbolic_bias, self.hyperbolic_input, self.nonlin = hyperbolic_bias, hyperbolic_input, nonlin
        self.k = torch.tensor(float(k))
        self.fp64_hyper = fp64_hyper

    def forward(self, input):
        return mobius_linear(input.float(), weight=self.weight, bias=self.bias, hyperbolic_input=self.hyperbolic_input,
                             nonlin=self.nonlin, hyperbolic_bias=self.hyperbolic_bias, k=float(self.k))
