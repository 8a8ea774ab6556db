"""Pair-wise Poincare distance (reference: hyperspace/poincare_distance.py:5-16) as one MFMA kernel."""
import torch

from .. import _C


def poincare_distance(pred, gt):
    """(N, D), (M, D) -> (N, M): acosh(1 + 2 clamp(|x-y|^2, 1e-7) / ((1-clamp(|x|^2,1e-5))(1-clamp(|y|^2,1e-5))))."""
    pred = _C.require_cuda(pred.to(torch.float32).contiguous(), "pred")
    gt = _C.require_cuda(gt.to(torch.float32).contiguous(), "gt")
    (n, d), (m, d2) = pred.shape, gt.shape
    if d != d2:
        raise _C.HypadError("poincare_distance: feature sizes differ")
    out = torch.empty(n, m, device=pred.device, dtype=torch.float32)
    _C.check(_C.lib.hypad_poincare_pairdist_fwd(_C.ptr(pred), _C.ptr(gt), _C.ptr(out), n, m, d, _C.stream()), "pairdist")
    return out
