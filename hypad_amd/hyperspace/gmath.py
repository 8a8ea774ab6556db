"""Poincare-ball ops (k = -1) on the GPU, with autograd.

Drop-in for the geoopt functions the reference's hot path reaches through
``geoopt.manifolds.stereographic.math`` (vendored at /root/reference/math_.py): expmap0 (:1132-1136),
logmap0 (:1267-1270), mobius_add (:536-555), project (:340-352), plus the inline row-wise distance of
train.py:226-230 and the hyperbolic loss of train.py:232.  Each is one HIP kernel forward and one backward.
"""
import torch

from .. import _C

_K_MSG = "only curvature k = -1 is implemented (the only value HypAD uses: hyperspace/hyrnn_nets.py:20,166)"


def _check_k(k):
    if k is not None and abs(float(k) + 1.0) > 1e-12:
        raise NotImplementedError(_K_MSG)


def _rows(t):
    t = t.to(torch.float32).contiguous()
    _C.require_cuda(t)
    return t, t.reshape(-1, t.shape[-1])


class _Unary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        x, x2 = _rows(x)
        out = torch.empty_like(x2)
        _C.check(getattr(_C.lib, fwd)(_C.ptr(x2), _C.ptr(out), x2.shape[0], x2.shape[1], _C.stream()), fwd)
        ctx.save_for_backward(x2)
        ctx.bwd, ctx.shape = bwd, x.shape
        return out.view(x.shape)

    @staticmethod
    @_C.first_order_only
    def backward(ctx, go):
        (x2,) = ctx.saved_tensors
        go2 = go.to(torch.float32).contiguous().reshape(x2.shape)
        gx = torch.empty_like(x2)
        _C.check(getattr(_C.lib, ctx.bwd)(_C.ptr(x2), _C.ptr(go2), _C.ptr(gx), x2.shape[0], x2.shape[1], _C.stream()), ctx.bwd)
        return gx.view(ctx.shape), None, None


def expmap0(u, *, k=None, dim=-1):
    _check_k(k)
    return _Unary.apply(u, "hypad_expmap0_fwd", "hypad_expmap0_bwd")


def logmap0(y, *, k=None, dim=-1):
    _check_k(k)
    return _Unary.apply(y, "hypad_logmap0_fwd", "hypad_logmap0_bwd")


def project(x, *, k=None, dim=-1, eps=-1.0):
    _check_k(k)
    if eps >= 0:
        raise NotImplementedError("custom eps: the fused kernel uses the fp32 default 4e-3 (math_.py:343-347)")
    return _Unary.apply(x, "hypad_project_fwd", "hypad_project_bwd")


class _MobiusAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, x2 = _rows(x)
        y, y2 = _rows(y)
        yr = y2.shape[0]
        if yr not in (1, x2.shape[0]) or y2.shape[1] != x2.shape[1]:
            raise _C.HypadError("mobius_add: y must have the rows of x or a single (broadcast) row")
        out = torch.empty_like(x2)
        _C.check(_C.lib.hypad_mobius_add_fwd(_C.ptr(x2), _C.ptr(y2), _C.ptr(out), x2.shape[0], x2.shape[1], yr, _C.stream()))
        ctx.save_for_backward(x2, y2)
        ctx.xs, ctx.ys = x.shape, y.shape
        return out.view(x.shape)

    @staticmethod
    @_C.first_order_only
    def backward(ctx, go):
        x2, y2 = ctx.saved_tensors
        go2 = go.to(torch.float32).contiguous().reshape(x2.shape)
        gx, gy = torch.empty_like(x2), torch.empty_like(x2)
        _C.check(_C.lib.hypad_mobius_add_bwd(_C.ptr(x2), _C.ptr(y2), _C.ptr(go2), _C.ptr(gx), _C.ptr(gy), x2.shape[0],
                                             x2.shape[1], y2.shape[0], _C.stream()))
        if y2.shape[0] == 1 and x2.shape[0] != 1:
            red = torch.empty(1, x2.shape[1], device=x2.device, dtype=torch.float32)
            _C.check(_C.lib.hypad_column_sum(_C.ptr(gy), _C.ptr(red), x2.shape[0], x2.shape[1], _C.stream()))
            gy = red
        return gx.view(ctx.xs), gy.view(ctx.ys)


def mobius_add(x, y, *, k=None, dim=-1):
    _check_k(k)
    if y.dim() == x.dim() and y.shape != x.shape:
        y = y.expand_as(x)
    if y.dim() < x.dim():
        y = y.reshape(1, -1)
    return _MobiusAdd.apply(x, y)


class _RowDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, v):
        u, u2 = _rows(u)
        v, v2 = _rows(v)
        out = torch.empty(u2.shape[0], device=u2.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_poincare_rowdist_fwd(_C.ptr(u2), _C.ptr(v2), _C.ptr(out), u2.shape[0], u2.shape[1], _C.stream()))
        ctx.save_for_backward(u2, v2)
        ctx.us, ctx.vs = u.shape, v.shape
        return out.view(u.shape[:-1])

    @staticmethod
    @_C.first_order_only
    def backward(ctx, gd):
        u2, v2 = ctx.saved_tensors
        gd = gd.to(torch.float32).contiguous().reshape(-1)
        gu, gv = torch.empty_like(u2), torch.empty_like(v2)
        _C.check(_C.lib.hypad_poincare_rowdist_bwd(_C.ptr(u2), _C.ptr(v2), _C.ptr(gd), _C.ptr(gu), _C.ptr(gv), u2.shape[0],
                                                   u2.shape[1], _C.stream()))
        return gu.view(ctx.us), gv.view(ctx.vs)


def poincare_rowdist(u, v):
    """acosh(1 + 2|u-v|^2 / ((1-|u|^2)(1-|v|^2)) + 1e-7) per row (train.py:226-230)."""
    return _RowDist.apply(u, v)


class _HyperLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, v, batch):
        u, u2 = _rows(u)
        v, v2 = _rows(v)
        out = torch.empty(1, device=u2.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_hyper_loss_fwd(_C.ptr(u2), _C.ptr(v2), _C.ptr(out), u2.shape[0], u2.shape[1], int(batch), _C.stream()))
        ctx.save_for_backward(u2, v2)
        ctx.us, ctx.vs, ctx.batch = u.shape, v.shape, int(batch)
        return out.view(())

    @staticmethod
    @_C.first_order_only
    def backward(ctx, g):
        u2, v2 = ctx.saved_tensors
        gu, gv = torch.empty_like(u2), torch.empty_like(v2)
        _C.check(_C.lib.hypad_hyper_loss_bwd(_C.ptr(u2), _C.ptr(v2), float(g), _C.ptr(gu), _C.ptr(gv), u2.shape[0], u2.shape[1],
                                             ctx.batch, _C.stream()))
        return gu.view(ctx.us), gv.view(ctx.vs), None


def hyperbolic_loss(u, v, batch_size):
    """torch.div(torch.sum(dist), batch_size)  (train.py:232)."""
    return _HyperLoss.apply(u, v, batch_size)
