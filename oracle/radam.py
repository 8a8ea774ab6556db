"""Riemannian Adam, restated (oracle; test infrastructure).  PARITY UNPINNED.

The reference calls ``geoopt.optim.RiemannianAdam(params, lr=..., weight_decay=1e-5,
stabilize=10)`` at ``train.py:282-288``.  geoopt==0.5.0 (``environment.yml:91``)
is neither installed in this image nor vendored in the reference tree (only its
math primitives are, as ``/root/reference/math_.py``), and the reference holds
no test or golden vector for it.  The update rule below restates geoopt 0.5.0's
published ``RiemannianAdam.step`` / ``Stereographic.retr_transp`` /
``Manifold.component_inner`` semantics:

  per param group:  step += 1
  per tensor p with gradient g:
      g   <- g + weight_decay * p
      g   <- egrad2rgrad(p, g)                 (Euclidean: identity; ball: g / lambda_p^2)
      m   <- b1 m + (1-b1) g
      v   <- b2 v + (1-b2) component_inner(p, g)
               (Euclidean: g*g; ball: lambda_p^2 <g,g>, one scalar broadcast over the vector)
      den <- sqrt(v / (1-b2^t)) + eps
      dir <- (m / (1-b1^t)) / den
      p'  <- retr(p, -lr dir)                  (Euclidean: p - lr dir; ball: project(p - lr dir))
      m   <- transp(p, p', m)                  (Euclidean: m; ball: gyr[p', -p] m * lambda_p / lambda_p')
  every ``stabilize`` steps: p <- project(p) for ball-valued tensors (proju is the identity).

The Euclidean branch is checked in tests/test_oracle_pins.py against
``torch.optim.Adam(weight_decay=...)`` (same L2-into-gradient rule).  Only the
ball branch (one 100-element bias, ``hyperbolic_linear.bias``) is unpinned.
"""
import torch

from . import gmath


def _is_ball(p) -> bool:
    return getattr(p, "manifold", None) is not None


class RiemannianAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, stabilize=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, stabilize=stabilize)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            group["step"] = group.get("step", 0) + 1
            t = group["step"]
            b1, b2 = group["betas"]
            lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                g = p.grad
                g.add_(p, alpha=wd)
                if _is_ball(p):
                    g = gmath.egrad2rgrad(p, g)
                    comp = gmath.inner(p, g, g, keepdim=True).expand_as(p)
                else:
                    comp = g * g
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).add_(comp, alpha=1 - b2)
                den = v.div(1 - b2 ** t).sqrt_().add_(eps)
                direction = m.div(1 - b1 ** t) / den
                if _is_ball(p):
                    new_p = gmath.project(p - lr * direction)
                    new_m = gmath.parallel_transport(p, new_p, m)
                    p.copy_(new_p)
                    m.copy_(new_m)
                else:
                    p.add_(direction, alpha=-lr)
            if group["stabilize"] is not None and t % group["stabilize"] == 0:
                for p in group["params"]:
                    if _is_ball(p) and len(self.state[p]) > 0:
                        p.copy_(gmath.project(p))
        return loss
