"""The mirrored networks are differentiable where the reference's are: a caller's own loss.backward() / optimizer.step() outside
the fused iteration functions (models/tadgan.py forwards under autograd).  Gradients of every parameter and of the input against
the CPU oracle modules (plain torch autograd on the reference's module structure)."""
import numpy as np
import pytest
import torch

from helpers import maxdiff

pytestmark = pytest.mark.gpu


def _pair(S, hyper, seed=5):
    from hypad_amd.models import tadgan
    from oracle import tadgan as ot
    torch.manual_seed(seed)
    o = dict(enc=ot.Encoder(S, 20), dec=ot.Decoder(S, 20, hyper), cx=ot.CriticX(S, 20), cz=ot.CriticZ(20))
    if hyper:
        with torch.no_grad():
            o["dec"].hyperbolic_linear.weight.mul_(50)
    h = dict(enc=tadgan.Encoder(S, 20), dec=tadgan.Decoder(S, 20, hyper), cx=tadgan.CriticX(S, 20), cz=tadgan.CriticZ(20))
    for k in o:
        h[k].load_state_dict(o[k].state_dict())
        h[k].cuda().eval()
        o[k].eval()                                   # dropout off on both sides; autograd still records
    return o, h


@pytest.mark.parametrize("S,hyper", [(100, True), (100, False), (51, True)])
def test_module_forwards_are_differentiable(S, hyper):
    o, h = _pair(S, hyper)
    rng = np.random.default_rng(S)
    B = 48
    x = rng.uniform(-1, 1, (B, S)).astype(np.float32)
    z = rng.standard_normal((B, 20)).astype(np.float32)
    wx, wz = rng.standard_normal((B, S)).astype(np.float32), rng.standard_normal((B, 20)).astype(np.float32)

    def loss_of(m, dev):
        t = lambda a: torch.from_numpy(a).to(dev)
        xi, zi = t(x).requires_grad_(True), t(z).requires_grad_(True)
        lat = m["enc"](xi.view(1, B, S))                                     # (1, B, 20)
        out = m["dec"](lat)
        rec = out[0] if hyper else out
        gen = m["dec"](zi.view(1, B, 20))
        gen = gen[0] if hyper else gen
        loss = (rec.view(B, S) * t(wx)).sum() + 0.3 * (lat.view(B, 20) * t(wz)).sum() + m["cx"](gen).sum() - 2.0 * m["cz"](lat).mean() \
            + (m["cx"](xi.view(1, B, S)) ** 2).mean()
        loss.backward()
        return loss, xi.grad, zi.grad

    lo, gxo, gzo = loss_of(o, "cpu")
    lh, gxh, gzh = loss_of(h, "cuda")
    assert abs(float(lh) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))
    assert maxdiff(gxh.cpu(), gxo) < 2e-5 * max(1.0, float(gxo.abs().max())) and maxdiff(gzh.cpu(), gzo) < 2e-5 * max(1.0, float(gzo.abs().max()))
    for k in o:
        ref = dict(o[k].named_parameters())
        for name, p in h[k].named_parameters():
            g = ref[name].grad
            assert p.grad is not None, (k, name)
            assert maxdiff(p.grad.cpu(), g) < 3e-5 * max(1.0, float(g.abs().max())), (k, name, maxdiff(p.grad.cpu(), g))
    assert float(h["enc"].lstm.weight_hh_l0.grad.abs().max()) == 0.0          # as in the reference


def test_custom_training_loop_outside_the_fused_iterations():
    """A reference user's own loop: forward, loss, backward, torch.optim.Adam.step() on the modules' parameters; the fused
    inference kernel (no_grad) then sees the updated weights (the parameters are views of the arena the kernels read)."""
    _, h = _pair(100, False, seed=9)
    enc, dec = h["enc"].train(), h["dec"].train()
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()), lr=2e-3)
    g = torch.Generator(device="cuda").manual_seed(0)
    t = torch.arange(64 + 99, device="cuda", dtype=torch.float32)
    series = torch.sin(t / 9.0)
    x = series.unfold(0, 100, 1)[:64].contiguous()
    first = last = None
    for step in range(60):
        opt.zero_grad()
        rec = dec(enc(x.view(1, 64, 100)))
        loss = torch.nn.functional.mse_loss(rec.view(64, 100), x)
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
        last = float(loss)
    assert last < 0.5 * first, (first, last)
    with torch.no_grad():
        rec = dec.eval()(enc.eval()(x.view(1, 64, 100)))                       # fused inference kernels, same arena
    assert abs(float(torch.nn.functional.mse_loss(rec.view(64, 100), x)) - last) < 0.2 * first


def test_double_backward_through_the_module_forwards_raises():
    """The layer Functions are first-order: a reference-style gradient penalty -- torch.autograd.grad(..., create_graph=True),
    train.py:72-93 -- must fail loudly instead of returning a constant whose second-order gradient is silently zero.  (The
    fused critic iterations carry the second-order chain themselves.)  Plain first-order use keeps working."""
    from hypad_amd import _C
    from hypad_amd.hyperspace import gmath
    o, h = _pair(100, True)
    x = torch.rand(32, 100, device="cuda").requires_grad_(True)
    out = h["cx"](x.view(1, 32, 100)).sum()
    with pytest.raises(_C.HypadError, match="create_graph"):
        torch.autograd.grad(out, x, create_graph=True)
    out = h["cx"](x.view(1, 32, 100)).sum()
    (g,) = torch.autograd.grad(out, x)
    assert g.shape == x.shape and bool(torch.isfinite(g).all())
    y = (0.3 * torch.rand(8, 100, device="cuda")).requires_grad_(True)
    with pytest.raises(_C.HypadError, match="create_graph"):
        torch.autograd.grad(gmath.expmap0(y).sum(), y, create_graph=True)
    with pytest.raises(_C.HypadError, match="create_graph"):
        torch.autograd.grad(h["dec"](torch.randn(1, 8, 20, device="cuda").requires_grad_(True))[0].sum(), list(h["dec"].parameters())[:1],
                            create_graph=True)
