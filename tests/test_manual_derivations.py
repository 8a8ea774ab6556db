"""The hand-derived backward formulas the HIP kernels transcribe (oracle/manual.py), checked on CPU:
(1) against the reference-generated gradient fixtures (eval mode, fp32), and
(2) against autograd of the same closed-form forward in fp64 with injected dropout masks."""
import numpy as np
import pytest
import torch

from helpers import load, maxdiff
from oracle import manual

torch.set_num_threads(1)


def _sd(fx, wkey, dtype=torch.float32):
    pre = wkey + "."
    return {k[len(pre):]: torch.from_numpy(np.array(v)).to(dtype) for k, v in fx.items() if k.startswith(pre)}


def _cmp(grads, fx, tag, tol=3e-6, wd=0.0):
    """wd: geoopt's RiemannianAdam adds weight_decay * p to p.grad IN PLACE (oracle/radam.py), so the
    gradients recorded after a hyperbolic decoder_iteration carry that term."""
    fx = dict(fx)
    if wd:
        for k in list(fx):
            if k.startswith(f"g1.{tag}."):
                fx[k] = fx[k] - wd * fx["w0." + k[len(f"g1.{tag}."):]]
    for k, g in grads.items():
        ref = fx[f"g1.{tag}.{k}"]
        assert maxdiff(g, ref) < tol * max(1.0, float(np.abs(ref).max())), (tag, k, maxdiff(g, ref))
    # tensors the manual path never touches must have exactly zero reference gradient (f gate rows, W_hh)
    for k, ref in fx.items():
        if k.startswith(f"g1.{tag}.") and k[len(f"g1.{tag}."):] not in grads:
            assert float(np.abs(ref).max()) == 0.0, k


@pytest.mark.parametrize("tag,hyper", [("hyper_S100", True), ("eucl_S100", False)])
def test_manual_matches_reference_fixtures(tag, hyper):
    fx = load(f"iters_{tag}.npz")
    sd = _sd(fx, "w0")
    x = torch.from_numpy(fx["samples"][0][:, :, 0]).float()
    with torch.no_grad():
        loss, g = manual.cx_iteration(sd, x, torch.from_numpy(fx["z_cx"][0]).float(),
                                      torch.from_numpy(fx["a_cx"][0]), hyper)
        assert abs(float(loss) - fx["loss_cx"][0]) < 1e-5
        _cmp(g, fx, "cx_iter")
        loss, g = manual.cz_iteration(sd, x, torch.from_numpy(fx["z_cz"][0]).float(), torch.from_numpy(fx["a_cz"][0]))
        assert abs(float(loss) - fx["loss_cz"][0]) < 1e-5
        _cmp(g, fx, "cz_iter")
        # decoder_iteration ran after `steps` critic updates in the fixture: use the trained critics
        sd2 = dict(sd)
        sd2.update({k: v for k, v in _sd(fx, "wN").items() if k.startswith(("cx.", "cz."))})
        loss, aux, g = manual.dec_iteration(sd2, x, torch.from_numpy(fx["z_dec"][0]).float(), hyper)
        assert abs(float(loss) - fx["loss_dec"][0]) < 2e-5
        assert abs(float(aux) - (fx["loss_hyper"][0] if hyper else fx["loss_mse"][0])) < 2e-5
        _cmp(g, fx, "dec_iter", tol=5e-6, wd=1e-5 if hyper else 0.0)
        # zero-gradient complement: f-gate rows of every weight_ih
        for k, v in g.items():
            if "weight_ih" in k:
                H = v.shape[0] // 4
                assert float(v[H:2 * H].abs().max()) == 0.0


def _rand_masks(gen, B, p, n, width=20):
    return [(torch.rand(B, width, generator=gen, dtype=torch.float64) >= p).double() / (1 - p) for _ in range(n)]


@pytest.mark.parametrize("hyper", [True, False])
def test_manual_backward_equals_autograd_fp64_with_dropout(hyper):
    fx = load("iters_hyper_S100.npz")
    sd = _sd(fx, "w0", torch.float64)
    # push the head towards the regime where every term matters
    sd["dec.hyperbolic_linear.weight"] = sd["dec.hyperbolic_linear.weight"] * 300
    sd["dec.hyperbolic_linear.bias"] = sd["dec.hyperbolic_linear.bias"] * 40
    gen = torch.Generator().manual_seed(3)
    B, S = 64, 100
    x = torch.from_numpy(fx["samples"][1][:, :, 0])
    z = torch.randn(B, 20, generator=gen, dtype=torch.float64)
    a_x = torch.rand(B, S, generator=gen, dtype=torch.float64)
    a_z = torch.rand(B, 20, generator=gen, dtype=torch.float64)
    dm = lambda: (torch.rand(B, 128, generator=gen, dtype=torch.float64) >= 0.2).double() / 0.8

    def leaf(keys):
        out = dict(sd)
        for k in sd:
            if k.startswith(keys):
                out[k] = sd[k].clone().requires_grad_(True)
        return out

    # ---- critic_x
    masks = dict(valid=_rand_masks(gen, B, .25, 4), fake=_rand_masks(gen, B, .25, 4),
                 inter=_rand_masks(gen, B, .25, 4), dec=dm())
    with torch.no_grad():
        loss_m, g_m = manual.cx_iteration(sd, x, z, a_x, hyper, masks)
    s = leaf(("cx.",))
    layers = manual.critic_layers(s, "cx.")
    valid, _, _ = manual.critic_fwd(x, layers, masks["valid"])
    genx = manual.decoder_fwd(z, s, hyper, masks["dec"])[0].detach()
    fake, _, _ = manual.critic_fwd(genx, layers, masks["fake"])
    inter = (a_x * x + (1 - a_x) * genx).requires_grad_(True)
    prob, _, _ = manual.critic_fwd(inter, layers, masks["inter"])
    gr = torch.autograd.grad(prob, inter, torch.ones_like(prob), create_graph=True)[0]
    gp = (torch.sqrt((gr ** 2).sum() + 1e-12) - 1) ** 2
    loss = fake.mean() - valid.mean() + 10 * gp
    loss.backward()
    assert abs(float(loss) - float(loss_m)) < 1e-12
    for k, g in g_m.items():
        assert maxdiff(g, s[k].grad) < 1e-11, k

    # ---- critic_z
    masks = dict(valid=_rand_masks(gen, B, .2, 2), fake=_rand_masks(gen, B, .2, 2), inter=_rand_masks(gen, B, .2, 2))
    with torch.no_grad():
        loss_m, g_m = manual.cz_iteration(sd, x, z, a_z, masks)
    s = leaf(("cz.",))
    layers = manual.critic_layers(s, "cz.")
    z_enc = manual.encoder_fwd(x, s)[0].detach()
    fake, _, _ = manual.critic_fwd(z_enc, layers, masks["fake"])
    valid, _, _ = manual.critic_fwd(z, layers, masks["valid"])
    inter = (a_z * z + (1 - a_z) * z_enc).requires_grad_(True)
    prob, _, _ = manual.critic_fwd(inter, layers, masks["inter"])
    gr = torch.autograd.grad(prob, inter, torch.ones_like(prob), create_graph=True)[0]
    loss = fake.mean() - valid.mean() + 10 * (torch.sqrt((gr ** 2).sum() + 1e-12) - 1) ** 2
    loss.backward()
    assert abs(float(loss) - float(loss_m)) < 1e-12
    for k, g in g_m.items():
        assert maxdiff(g, s[k].grad) < 1e-11, k

    # ---- decoder/encoder
    masks = dict(cz=_rand_masks(gen, B, .2, 2), cx=_rand_masks(gen, B, .25, 4), dec_gen=dm(), dec_rec=dm())
    with torch.no_grad():
        loss_m, aux_m, g_m = manual.dec_iteration(sd, x, z, hyper, masks)
    s = leaf(("dec.", "enc."))
    z_enc, _ = manual.encoder_fwd(x, s)
    fake_z, _, _ = manual.critic_fwd(z_enc, manual.critic_layers(s, "cz."), masks["cz"])
    genx = manual.decoder_fwd(z, s, hyper, masks["dec_gen"])[0]
    fake_x, _, _ = manual.critic_fwd(genx, manual.critic_layers(s, "cx."), masks["cx"])
    rec = manual.decoder_fwd(z_enc, s, hyper, masks["dec_rec"])[0]
    if hyper:
        hx, _ = manual.head_fwd(x, s)
        aux = manual.rowdist_fwd(rec, hx).sum() / B
    else:
        aux = ((rec - x) ** 2).mean()
    loss = 10 * aux - fake_x.mean() - fake_z.mean()
    loss.backward()
    assert abs(float(loss) - float(loss_m)) < 1e-11 and abs(float(aux) - float(aux_m)) < 1e-12
    for k, v in s.items():
        if v.requires_grad and v.grad is not None and (hyper or "hyperbolic_linear" not in k):
            got = g_m.get(k)
            if got is None:
                assert float(v.grad.abs().max()) == 0.0, k      # W_hh
            else:
                assert maxdiff(got, v.grad) < 1e-10 * max(1.0, float(v.grad.abs().max())), k


def test_head_backward_edge_rows_fp64():
    """clipped rows (project active), tanh-saturated rows and near-zero rows."""
    torch.manual_seed(0)
    S = 100
    b = manual.head_epilogue_fwd(torch.randn(1, S, dtype=torch.float64) / 12, torch.zeros(S, dtype=torch.float64))[0]
    u = torch.randn(12, S, dtype=torch.float64)
    u = u / u.norm(dim=1, keepdim=True) * torch.tensor([1e-9, 1e-3, 0.1, 0.5, 1.0, 2.0, 3.0, 3.5, 5.0, 9.0, 14.0, 20.0],
                                                        dtype=torch.float64).unsqueeze(1)
    u.requires_grad_(True)
    bb = b.clone().requires_grad_(True)
    r = manual.head_epilogue_fwd(u, bb)
    dr = torch.randn(12, S, dtype=torch.float64)
    gu, gb = torch.autograd.grad(r, (u, bb), dr)
    with torch.no_grad():
        du, db = manual.head_epilogue_bwd(u, bb, dr)
    assert maxdiff(du, gu) < 1e-10 and maxdiff(db.sum(0), gb) < 1e-10
    assert float((r.norm(dim=1) > 0.9959).sum()) >= 3      # the clipped branch really ran


def test_manual_optimizer_rules():
    from oracle.radam import RiemannianAdam
    from oracle.tadgan import BallParameter
    torch.manual_seed(0)
    p0, lr = torch.randn(50), 5e-4
    a = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([a], lr=lr)
    p, m, v = p0.clone(), torch.zeros(50), torch.zeros(50)
    for t in range(1, 13):
        g = torch.randn(50)
        a.grad = g.clone(); opt.step()
        p, m, v = manual.adam_step(p, g, m, v, t, lr)
    assert maxdiff(p, a.detach()) < 1e-6
    # ball branch vs oracle.radam
    b0 = manual.head_epilogue_fwd(torch.randn(1, 100) / 6, torch.zeros(100))[0]
    bp = BallParameter(b0.clone())
    ro = RiemannianAdam([bp], lr=lr, weight_decay=1e-5, stabilize=10)
    p, m, v = b0.clone(), torch.zeros(100), torch.zeros(100)
    for t in range(1, 23):
        g = torch.randn(100) * 0.1
        bp.grad = g.clone(); ro.step()
        p, m, v = manual.radam_ball_step(p, g, m, v, t, lr)
    assert maxdiff(p, bp.detach()) < 1e-6
    assert maxdiff(m, ro.state[bp]["exp_avg"]) < 1e-6 and maxdiff(v, ro.state[bp]["exp_avg_sq"]) < 1e-6
