"""Optimizers of the hot path as stand-alone HIP steps.

``Adam`` follows ``torch.optim.Adam`` (train.py:274-281); ``RiemannianAdam`` restates
``geoopt.optim.RiemannianAdam`` (train.py:282-288; geoopt==0.5.0 -- not vendored in the reference, see
oracle/radam.py for the rule and its pinning status).  Inside the fused training iterations the same update
runs in the epilogue of the weight-gradient kernel; these classes exist for callers that bring their own
``.grad`` tensors, and to carry hyper-parameters / state for ``hypad_amd.train``.
"""
import torch

from . import _C


def _is_ball(p):
    return getattr(p, "manifold", None) is not None


class _Base(torch.optim.Optimizer):
    riemannian = False

    def _state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            group["step"] = group.get("step", 0) + 1
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _C.require_cuda(p.data, "parameter")
                g = _C.require_cuda(p.grad.contiguous(), "gradient")
                st = self._state(p)
                st["step"] = group["step"]
                n = p.numel()
                if self.riemannian:
                    ball = n if _is_ball(p) else 0
                    rc = _C.lib.hypad_radam_step(_C.ptr(p.data), _C.ptr(g), _C.ptr(st["exp_avg"]), _C.ptr(st["exp_avg_sq"]), n, 0,
                                                 ball, group["step"], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                                 int(group.get("stabilize") or 0), _C.stream())
                else:
                    rc = _C.lib.hypad_adam_step(_C.ptr(p.data), _C.ptr(g), _C.ptr(st["exp_avg"]), _C.ptr(st["exp_avg_sq"]), n,
                                                group["step"], group["lr"], b1, b2, group["eps"], group["weight_decay"], _C.stream())
                _C.check(rc, type(self).__name__ + ".step")
        return loss


class Adam(_Base):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))


class RiemannianAdam(_Base):
    riemannian = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, stabilize=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, stabilize=stabilize))
