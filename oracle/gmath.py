"""Poincare-ball primitives, restated for curvature k = -1 (oracle; test infrastructure).

The reference reaches these through ``geoopt.manifolds.stereographic.math``
(geoopt==0.5.0); a verbatim copy of that module sits at ``/root/reference/math_.py``
and is the source followed here.  Every function below is specialised to the
only curvature the hot path uses (``k = -1``: ``hyperspace/hyrnn_nets.py:20,166``)
and keeps the reference's clamp constants.  All are plain torch
ops, so autograd supplies the oracle for the hand-written backward kernels.
"""
import torch

MIN_NORM = 1e-15          # math_.py:1134,1269,349
TANH_CLAMP = 15.0         # math_.py:53
ARTANH_EPS = 1e-7         # math_.py:58
PROJ_EPS_F32 = 4e-3       # math_.py:343-347
PROJ_EPS_F64 = 1e-5
SABS_EPS = 1e-15          # geoopt.utils.sabs, used by tan_k/artan_k (math_.py:226,250)


def _sqrt_abs_k(ref: torch.Tensor) -> torch.Tensor:
    # sabs(k).sqrt() with k = -1 (math_.py:226): sqrt(1 + 1e-15) == 1 in fp32 and fp64.
    return torch.sqrt(torch.ones((), dtype=ref.dtype) + SABS_EPS)


def tanh_clamped(x):
    """math_.py:51-53"""
    return x.clamp(-TANH_CLAMP, TANH_CLAMP).tanh()


def artanh(x):
    """math_.py:56-59"""
    x = x.clamp(-1 + ARTANH_EPS, 1 - ARTANH_EPS)
    return (torch.log(1 + x) - torch.log(1 - x)) * 0.5


def expmap0(u):
    """math_.py:1132-1136 with tan_k (math_.py:217-238) at k=-1."""
    ks = _sqrt_abs_k(u)
    n = u.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    return (tanh_clamped(n * ks) / ks) * (u / n)


def logmap0(y):
    """math_.py:1267-1270 with artan_k (math_.py:241-262) at k=-1."""
    ks = _sqrt_abs_k(y)
    n = y.norm(dim=-1, p=2, keepdim=True).clamp_min(MIN_NORM)
    return (y / n) * (artanh(n * ks) / ks)


def mobius_add(x, y):
    """math_.py:536-555 at k=-1."""
    x2 = x.pow(2).sum(dim=-1, keepdim=True)
    y2 = y.pow(2).sum(dim=-1, keepdim=True)
    xy = (x * y).sum(dim=-1, keepdim=True)
    num = (1 + 2 * xy + y2) * x + (1 - x2) * y
    den = 1 + 2 * xy + x2 * y2
    return num / den.clamp_min(MIN_NORM)


def project(x, eps: float = -1.0):
    """math_.py:340-352 at k=-1."""
    if eps < 0:
        eps = PROJ_EPS_F32 if x.dtype == torch.float32 else PROJ_EPS_F64
    maxnorm = (1 - eps) / _sqrt_abs_k(x)
    n = x.norm(dim=-1, keepdim=True, p=2).clamp_min(MIN_NORM)
    return torch.where(n > maxnorm, x / n * maxnorm, x)


def lambda_x(x, keepdim=False):
    """math_.py:382-383 at k=-1."""
    return 2 / (1 - x.pow(2).sum(dim=-1, keepdim=keepdim)).clamp_min(MIN_NORM)


def inner(x, u, v, keepdim=False):
    """math_.py:419-430."""
    return lambda_x(x, keepdim=True) ** 2 * (u * v).sum(dim=-1, keepdim=keepdim)


def egrad2rgrad(x, grad):
    """math_.py:1843-1845."""
    return grad / lambda_x(x, keepdim=True) ** 2


def gyration(u, v, w):
    """math_.py:656-676 at k=-1 (k**2 = 1)."""
    u2 = u.pow(2).sum(dim=-1, keepdim=True)
    v2 = v.pow(2).sum(dim=-1, keepdim=True)
    uv = (u * v).sum(dim=-1, keepdim=True)
    uw = (u * w).sum(dim=-1, keepdim=True)
    vw = (v * w).sum(dim=-1, keepdim=True)
    a = -uw * v2 + vw + 2 * uv * vw
    b = -vw * u2 - uw
    d = 1 + 2 * uv + u2 * v2
    return w + 2 * (a * u + b * v) / d.clamp_min(MIN_NORM)


def parallel_transport(x, y, u):
    """math_.py:1738-1746."""
    return gyration(y, -x, u) * lambda_x(x, keepdim=True) / lambda_x(y, keepdim=True)


def mobius_linear(inp, weight, bias):
    """hyperspace/hyrnn_nets.py:13-35 in the one configuration the hot path uses
    (hyperbolic_input=False, hyperbolic_bias=True, nonlin=None, k=-1;
    models/tadgan.py:43-52)."""
    out = torch.nn.functional.linear(inp, weight)
    out = expmap0(out)
    out = mobius_add(out, bias.unsqueeze(0).expand_as(out))
    return project(out)


def rowwise_poincare_distance(u, v):
    """train.py:226-230 and utils/anomaly_detection_utils.py:58-66,167-175."""
    sqdist = torch.sum((u - v) ** 2, dim=-1)
    squnorm = torch.sum(u ** 2, dim=-1)
    sqvnorm = torch.sum(v ** 2, dim=-1)
    return torch.acosh(1 + 2 * sqdist / ((1 - squnorm) * (1 - sqvnorm)) + 1e-7)


def pairwise_poincare_distance(pred, gt):
    """hyperspace/poincare_distance.py:5-16 (+ :19-25, :28-48)."""
    def sq_norm(x):
        return torch.clamp(torch.norm(x, dim=-1, p=2) ** 2, min=1e-5)

    a = (1 - sq_norm(pred)).view(-1, 1)
    b = (1 - sq_norm(gt)).view(1, -1)
    xn = (pred ** 2).sum(1).view(-1, 1)
    yn = (gt ** 2).sum(1).view(1, -1)
    d = torch.clamp(xn + yn - 2.0 * torch.mm(pred, gt.t()), 1e-7, float("inf"))
    return torch.acosh(1 + 2 * d / torch.matmul(a, b))
