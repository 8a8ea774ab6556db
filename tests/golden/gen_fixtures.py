#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):

    PYTORCH_JIT=0 PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_fixtures.py

Every array written is an input or an output of a reference function
(``models.tadgan``, ``hyperspace.*``, ``train.*_iteration``,
``utils.anomaly_detection_utils.*`` imported from /root/reference through
tests/golden/refharness.py) -- data only, no reference source.  The versions that
produced them are recorded in ``versions.json``.
"""
import json
import os
import sys
from types import SimpleNamespace

os.environ.setdefault("PYTORCH_JIT", "0")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import refharness  # noqa: E402

refharness.install()

import geoopt.manifolds.stereographic.math as gmath  # noqa: E402  (the reference's vendored math_.py)
import models.tadgan as ref_tadgan  # noqa: E402
import train as ref_train  # noqa: E402
from hyperspace.hyrnn_nets import mobius_linear as ref_mobius_linear  # noqa: E402
from hyperspace.poincare_distance import poincare_distance as ref_pairdist  # noqa: E402
import utils.anomaly_detection_utils as ref_adu  # noqa: E402

torch.set_num_threads(1)
K = torch.tensor(-1.0)
F32 = np.float32


def sd_np(prefix, module):
    return {f"{prefix}.{k}": v.detach().numpy().copy() for k, v in module.state_dict().items()}


def build(S, L=20, seed=0):
    torch.manual_seed(seed)
    enc = ref_tadgan.Encoder(S, L)
    dec = ref_tadgan.Decoder(S, L, True)
    cx = ref_tadgan.CriticX(S, L)
    cz = ref_tadgan.CriticZ(L)
    dec_e = ref_tadgan.Decoder(S, L, False)
    # Euclidean decoder shares every non-head tensor with the hyperbolic one (keeps the fixture small)
    dec_e.load_state_dict({k: v for k, v in dec.state_dict().items() if not k.startswith("hyperbolic_linear")})
    return enc, dec, dec_e, cx, cz


def synth_windows(n, S, seed=0):
    """SURVEY.md §8d config-1 stand-in: sine + noise + one square jump, clipped to [-1, 1], float64 (n,S,1)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n + S - 1)
    series = np.sin(2 * np.pi * t / 288.0) + 0.05 * rng.standard_normal(len(t))
    series[n // 2: n // 2 + 40] += 0.8
    series = np.clip(series, -1, 1)
    idx = np.arange(n)[:, None] + np.arange(S)[None, :]
    return series[idx][:, :, None].astype(np.float64)


# ------------------------------------------------------------------------------ weights + forward
def gen_forward(S, B, tag):
    enc, dec, dec_e, cx, cz = build(S)
    for m in (enc, dec, dec_e, cx, cz):
        m.eval()
    out = {}
    for p, m in (("enc", enc), ("dec", dec), ("cx", cx), ("cz", cz)):
        out.update(sd_np(p, m))
    rng = np.random.default_rng(100 + S)
    x = synth_windows(B, S, seed=1) if S == 100 else rng.uniform(-1, 1, size=(B, S, 1))
    z = rng.standard_normal((B, 20)).astype(F32)
    xt, zt = torch.from_numpy(x), torch.from_numpy(z).view(1, B, 20)
    with torch.no_grad():
        hyper, eucl = dec(zt)
        out.update(x=x, z=z,
                   enc_x=enc(xt).numpy(), dec_hyper=hyper.numpy(), dec_eucl=eucl.numpy(),
                   dec_e_out=dec_e(zt).numpy(),
                   head_x=dec.hyperbolic_linear(xt.view(-1, S).float()).numpy(),
                   cx_x=cx(xt).numpy(), cz_z=cz(zt).numpy())
        # test_tadgan batch body (anomaly_detection.py:67-95): enc -> dec -> head(sample) -> cx(sample)
        lat = enc(xt.float())
        h2, e2 = dec(lat)
        out.update(s0_hyper=h2.numpy(), s0_eucl=e2.numpy())
    np.savez(os.path.join(HERE, f"fwd_{tag}.npz"), **out)
    return enc, dec, dec_e, cx, cz


# ------------------------------------------------------------------------------ hyperbolic ops
def edge_rows(S, rng):
    rows = [np.zeros(S), np.full(S, 1e-20 / np.sqrt(S))]
    for nrm in (1e-8, 0.3, 0.9, 0.995, 0.9961, 0.999, 0.99999995, 1.5, 20.0):
        v = rng.standard_normal(S)
        rows.append(v / np.linalg.norm(v) * nrm)
    return np.asarray(rows, dtype=F32)


def gen_ops(S=100):
    rng = np.random.default_rng(7)
    out = {}
    u = np.concatenate([rng.standard_normal((40, S)).astype(F32) * s for s in (0.01, 0.1, 1.0)] + [edge_rows(S, rng)])
    ball = np.concatenate([
        (rng.standard_normal((40, S)) * 0.05).astype(F32),
        (lambda v: (v / np.linalg.norm(v, axis=1, keepdims=True) * rng.uniform(0.5, 0.99, (40, 1))).astype(F32))(
            rng.standard_normal((40, S))),
        edge_rows(S, rng)[:9]])          # keep inside (or barely outside) the ball
    bias = gmath.expmap0(torch.from_numpy(rng.standard_normal(S).astype(F32)) / 400, k=K).numpy()
    bias_big = gmath.expmap0(torch.from_numpy(rng.standard_normal(S).astype(F32)) / 12, k=K).numpy()
    y2 = np.concatenate([(rng.standard_normal((len(ball) - 9, S)) * 0.04).astype(F32), ball[:9][::-1]])

    def fwd_bwd(name, fn, *inputs):
        ts = [torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(True) for a in inputs]
        o = fn(*ts)
        go = torch.from_numpy(np.random.default_rng(len(name)).standard_normal(tuple(o.shape)).astype(F32))
        gs = torch.autograd.grad(o, ts, go, allow_unused=True)
        out[f"{name}_out"] = o.detach().numpy()
        out[f"{name}_gout"] = go.numpy()
        for i, g in enumerate(gs):
            out[f"{name}_gin{i}"] = g.numpy()

    out.update(u=u, ball=ball, bias=bias, bias_big=bias_big, y2=y2)
    fwd_bwd("expmap0", lambda a: gmath.expmap0(a, k=K), u)
    fwd_bwd("logmap0", lambda a: gmath.logmap0(a, k=K), ball)
    fwd_bwd("mobius_add", lambda a, b: gmath.mobius_add(a, b, k=K), ball, y2)
    fwd_bwd("mobius_add_bias", lambda a, b: gmath.mobius_add(a, b.unsqueeze(0).expand_as(a), k=K), ball, bias_big)
    fwd_bwd("project", lambda a: gmath.project(a, k=K), u)
    W = (rng.standard_normal((S, S)) * 0.02).astype(F32)
    out["W"] = W
    fwd_bwd("mobius_linear", lambda a, w, b: ref_mobius_linear(a, w, b, hyperbolic_input=False, hyperbolic_bias=True,
                                                                nonlin=None, k=-1.0), u[:120] * 0.5, W, bias_big)
    fwd_bwd("mobius_linear_small", lambda a, w, b: ref_mobius_linear(a, w, b, hyperbolic_input=False,
                                                                      hyperbolic_bias=True, nonlin=None, k=-1.0),
            u[:120] * 0.5, W * 0.01, bias)

    def rowdist(a, b):   # train.py:226-230
        sqdist = torch.sum((a - b) ** 2, dim=-1)
        return torch.acosh(1 + 2 * sqdist / ((1 - torch.sum(a ** 2, dim=-1)) * (1 - torch.sum(b ** 2, dim=-1))) + 1e-7)

    inside = ball[:80]
    fwd_bwd("rowdist", rowdist, inside, np.roll(inside, 3, axis=0) * 0.9)
    fwd_bwd("rowdist_same", rowdist, inside, inside.copy())
    pa = np.concatenate([inside[:30], inside[:2], np.zeros((2, S), F32)])
    pb = np.concatenate([inside[40:70] * 0.8, inside[:3]])
    fwd_bwd("pairdist", ref_pairdist, pa, pb)
    np.savez(os.path.join(HERE, "ops.npz"), **out)


# ------------------------------------------------------------------------------ training iterations
class _Feed:
    """Replaces numpy.random.normal / torch.rand inside the reference's iteration functions with
    recorded draws (host RNG)."""

    def __init__(self, zs, alphas):
        self.zs, self.alphas = list(zs), list(alphas)

    def __enter__(self):
        self._n, self._r = np.random.normal, torch.rand
        np.random.normal = lambda size=None, **k: self.zs.pop(0).reshape(size)
        torch.rand = lambda *shape, **k: self.alphas.pop(0).reshape(*shape)
        return self

    def __exit__(self, *a):
        np.random.normal, torch.rand = self._n, self._r


def grads_np(prefix, module):
    return {f"g.{prefix}.{k}": (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), F32))
            for k, p in module.named_parameters()}


def gen_iters(S, B, hyperbolic, tag, steps=6):
    enc, dec, dec_e, cx, cz = build(S)
    if not hyperbolic:
        dec = dec_e
    for m in (enc, dec, cx, cz):
        m.eval()                     # dropout off: bit-parity with CPU generators is impossible (SURVEY §7 hard part 4)
    params = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=20, lr=5e-4, hyperbolic=hyperbolic)
    rng = np.random.default_rng(11)
    data = synth_windows(B * steps, S, seed=3)
    perm = rng.permutation(len(data))
    samples = [torch.from_numpy(data[perm[i * B:(i + 1) * B]]) for i in range(steps)]
    out = dict(samples=np.stack([s.numpy() for s in samples]))
    for p, m in (("enc", enc), ("dec", dec), ("cx", cx), ("cz", cz)):
        out.update({f"w0.{k}": v for k, v in sd_np(p, m).items()})

    ocx = torch.optim.Adam(cx.parameters(), lr=params.lr, betas=(0.9, 0.999))
    ocz = torch.optim.Adam(cz.parameters(), lr=params.lr, betas=(0.9, 0.999))
    if hyperbolic:
        import geoopt
        odec = geoopt.optim.RiemannianAdam(list(dec.parameters()) + list(enc.parameters()), lr=params.lr,
                                           weight_decay=1e-5, stabilize=10)      # oracle.radam stand-in (UNPINNED)
    else:
        odec = torch.optim.Adam(list(dec.parameters()) + list(enc.parameters()), lr=params.lr, betas=(0.9, 0.999))

    z_cx = rng.standard_normal((steps, B, 20))
    z_cz = rng.standard_normal((steps, B, 20))
    z_dec = rng.standard_normal((steps, B, 20))
    a_cx = rng.uniform(size=(steps, B, S)).astype(F32)
    a_cz = rng.uniform(size=(steps, B, 20)).astype(F32)
    out.update(z_cx=z_cx, z_cz=z_cz, z_dec=z_dec, a_cx=a_cx, a_cz=a_cz)

    # critic phase (train.py:306-328): generator frozen
    ref_train_set(enc, dec, False)
    ref_train_set(cx, cz, True)
    l_cx, l_cz = [], []
    for i in range(steps):
        with _Feed([z_cx[i]], [torch.from_numpy(a_cx[i])]):
            l = ref_train.critic_x_iteration(samples[i], dec, cx, ocx, params)
        l_cx.append(float(l))
        if i == 0:
            out["cx_loss_dtype"] = np.array(str(l.dtype))
            out.update({k.replace("g.", "g1.cx_iter."): v for k, v in grads_np("cx", cx).items()})
            out.update({f"w1.{k}": v for k, v in sd_np("cx", cx).items()})
        with _Feed([z_cz[i]], [torch.from_numpy(a_cz[i])]):
            l = ref_train.critic_z_iteration(samples[i], enc, cz, ocz, params)
        l_cz.append(float(l))
        if i == 0:
            out.update({k.replace("g.", "g1.cz_iter."): v for k, v in grads_np("cz", cz).items()})
            out.update({f"w1.{k}": v for k, v in sd_np("cz", cz).items()})
    out.update(loss_cx=np.asarray(l_cx), loss_cz=np.asarray(l_cz))
    out.update({f"wN.{k}": v for k, v in sd_np("cx", cx).items()})
    out.update({f"wN.{k}": v for k, v in sd_np("cz", cz).items()})

    # generator phase (train.py:333-352): critics frozen at their trained state
    ref_train_set(enc, dec, True)
    ref_train_set(cx, cz, False)
    l_dec, l_hyp, l_mse = [], [], []
    for i in range(steps):
        with _Feed([z_dec[i]], []):
            l, h, m = ref_train.decoder_iteration(samples[i], enc, dec, cx, cz, odec, params)
        l_dec.append(float(l)); l_hyp.append(float(h)); l_mse.append(float(m))
        if i == 0:
            out.update({k.replace("g.", "g1.dec_iter."): v for k, v in grads_np("dec", dec).items()})
            out.update({k.replace("g.", "g1.dec_iter."): v for k, v in grads_np("enc", enc).items()})
            out.update({f"w1.{k}": v for k, v in sd_np("dec", dec).items()})
            out.update({f"w1.{k}": v for k, v in sd_np("enc", enc).items()})
    out.update(loss_dec=np.asarray(l_dec), loss_hyper=np.asarray(l_hyp), loss_mse=np.asarray(l_mse))
    out.update({f"wN.{k}": v for k, v in sd_np("dec", dec).items()})
    out.update({f"wN.{k}": v for k, v in sd_np("enc", enc).items()})
    np.savez(os.path.join(HERE, f"iters_{tag}.npz"), **out)


def ref_train_set(a, b, flag):
    for m in (a, b):
        for p in m.parameters():
            p.requires_grad = flag


# ------------------------------------------------------------------------------ scoring
def gen_scoring(N=300, S=100):
    rng = np.random.default_rng(5)
    y = synth_windows(N, S, seed=9)
    y_hat = (y[:, :, 0] + 0.05 * rng.standard_normal((N, S))).astype(F32)
    y_hat[N // 3: N // 3 + 20] += 0.4
    critic = rng.standard_normal(N).astype(F32)
    out = dict(y=y, y_hat=y_hat, critic=critic)
    w = int(N * 0.01)
    err, pvs = ref_adu.reconstruction_errors(y, y_hat, 1, 10, w, True, "point")
    out.update(point_err=np.asarray(err, dtype=np.float64), predictions_vs=np.asarray(pvs, dtype=np.float64))
    err_raw, _ = ref_adu.reconstruction_errors(y, y_hat, 1, 10, w, False, "point")
    out.update(point_err_raw=np.asarray(err_raw, dtype=np.float64))
    from scipy import stats
    out["point_z"] = np.clip(stats.zscore(err), a_min=0, a_max=None) + 1
    cs = ref_adu.final_critic_scores(list(critic), y)
    out["critic_scores"] = np.asarray(cs, dtype=np.float64)
    out["critic_score_direct"] = ref_adu._compute_critic_score(critic, 7)
    # hyperbolic branch of univariate_anomaly_detection (:54-86), minus I/O
    ball_a = (0.3 * np.tanh(y_hat)).astype(F32)
    ball_b = (0.3 * np.tanh(y[:, :, 0])).astype(F32)
    ta, tb = torch.Tensor(ball_a).reshape(-1, S), torch.Tensor(ball_b).reshape(-1, S)
    sqdist = torch.sum((tb - ta) ** 2, dim=1)
    rec = torch.acosh(1 + 2 * sqdist / ((1 - torch.sum(tb ** 2, dim=-1)) * (1 - torch.sum(ta ** 2, dim=-1))) + 1e-7)
    out.update(ball_recons=ball_a, ball_real=ball_b, hyper_rec=rec.numpy())
    crit = np.asarray(cs)[: rec.shape[0]]
    for comb in ("sum", "mult", "uncertainty", "critic", "critic_uncertainty", "sum_uncertainty", "rec",
                 "rec_uncertainty"):
        out[f"comb_{comb}"] = np.asarray(ref_adu.combine_scores(comb, crit, rec.numpy(), ball_a), dtype=np.float64)
    # Euclidean branch (score_anomalies :407-576, path=None so nothing is pickled), point error
    for comb in ("mult", "sum", "rec", "critic"):
        fs, _, true, _ = ref_adu.score_anomalies(y, y_hat, critic, None, rec_error_type="point", comb=comb)
        out[f"eucl_{comb}"] = np.asarray(fs, dtype=np.float64)
    out["true_unrolled"] = np.asarray(true, dtype=np.float64).reshape(-1)
    np.savez(os.path.join(HERE, "score.npz"), **out)



# ------------------------------------------------------------------------------ Riemannian primitives + ball Adam (row O2)
def gen_riemann(S=100, steps=23):
    """The Riemannian primitives geoopt's optimizer is assembled from, evaluated by the REFERENCE's vendored copy
    (math_.py:382-383 lambda_x, :419-430 inner, :656-676 gyration, :1738-1746 parallel_transport, :1843-1845
    egrad2rgrad, :340-352 project), plus trajectories of the one ball-valued parameter (`hyperbolic_linear.bias`)
    under geoopt 0.5.0's published ``RiemannianAdam.step`` ORDER, every arithmetic step of which is a call into
    that vendored module (no oracle code).  What stays unpinned after this fixture is only that published order."""
    rng = np.random.default_rng(23)
    out = {}
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def on_ball(n, lo, hi):
        v = rng.standard_normal((n, S))
        return (v / np.linalg.norm(v, axis=1, keepdims=True) * rng.uniform(lo, hi, (n, 1))).astype(F32)

    x = np.concatenate([on_ball(12, 0.0, 0.3), on_ball(12, 0.3, 0.95), on_ball(4, 0.99, 0.996),
                        np.zeros((1, S), F32), on_ball(1, 0.9999999, 1.0), on_ball(2, 1.0, 1.2)])   # incl. lambda's 1e-15 floor side
    y = np.concatenate([on_ball(16, 0.0, 0.5), on_ball(16, 0.5, 0.995)])
    u = (rng.standard_normal(x.shape) * rng.uniform(1e-3, 2.0, (len(x), 1))).astype(F32)
    v = (rng.standard_normal(x.shape) * 0.3).astype(F32)
    out.update(prim_x=x, prim_y=y, prim_u=u, prim_v=v)
    out["lambda_x"] = gmath.lambda_x(T(x), k=K, keepdim=True).numpy()
    out["inner"] = gmath.inner(T(x), T(u), T(v), k=K, keepdim=True).numpy()
    out["inner_uu"] = gmath.inner(T(x), T(u), T(u), k=K, keepdim=True).numpy()
    out["egrad2rgrad"] = gmath.egrad2rgrad(T(x), T(u), k=K).numpy()
    out["gyration"] = gmath.gyration(T(x), T(y), T(u), k=K).numpy()
    out["parallel_transport"] = gmath.parallel_transport(T(x), T(y), T(u), k=K).numpy()
    out["project_x"] = gmath.project(T(x), k=K).numpy()

    def trajectory(tag, p0, grads, lr, wd=1e-5, b1=0.9, b2=0.999, eps=1e-8, stabilize=10):
        p = T(p0).clone()
        m, vv = torch.zeros_like(p), torch.zeros_like(p)
        P, M, V = [], [], []
        for t in range(1, len(grads) + 1):
            g = T(grads[t - 1]).clone()
            g.add_(p, alpha=wd)                                              # weight decay into the gradient
            g = gmath.egrad2rgrad(p, g, k=K)                                 # math_.py:1843
            m.mul_(b1).add_(g, alpha=1 - b1)
            vv.mul_(b2).add_(gmath.inner(p, g, g, k=K, keepdim=True), alpha=1 - b2)   # component_inner: math_.py:419 broadcast
            denom = vv.div(1 - b2 ** t).sqrt_()
            direction = m.div(1 - b1 ** t) / denom.add_(eps)
            new_p = gmath.project(p + (-lr * direction), k=K)                # Stereographic.retr: math_.py:340
            new_m = gmath.parallel_transport(p, new_p, m, k=K)               # Stereographic.transp: math_.py:1738
            p.copy_(new_p); m.copy_(new_m)
            if stabilize is not None and t % stabilize == 0:
                p.copy_(gmath.project(p, k=K))                               # stabilize_group: projx (proju = identity)
            P.append(p.numpy().copy()); M.append(m.numpy().copy()); V.append(vv.numpy().copy())
        out.update({f"traj_{tag}_p0": p0, f"traj_{tag}_grads": np.asarray(grads), f"traj_{tag}_lr": np.float64(lr),
                    f"traj_{tag}_p": np.asarray(P), f"traj_{tag}_m": np.asarray(M), f"traj_{tag}_v": np.asarray(V)})

    # (a) the reference's own initialisation (hyperspace/hyrnn_nets.py:176-179) and learning rate (configs/univariate.yaml)
    p0 = gmath.expmap0(T(rng.standard_normal(S).astype(F32)) / 400, k=K).numpy()
    trajectory("init", p0, [(rng.standard_normal(S) * 0.1).astype(F32) for _ in range(steps)], 5e-4)
    # (b) mid-ball, larger steps: the gyration / conformal-factor ratio of the transport is far from the identity
    p0 = on_ball(1, 0.7, 0.7)[0]
    trajectory("mid", p0, [(rng.standard_normal(S) * 2.0).astype(F32) for _ in range(steps)], 2e-2)
    # (c) against the boundary: gradients push outwards, so the retraction's projection (norm > 1 - 4e-3) fires
    #     (the Riemannian step has length ~ lr / lambda_p, lambda_p ~ 230 here, hence the start at 0.9955 and lr = 0.05)
    p0 = on_ball(1, 0.9955, 0.9955)[0]
    trajectory("edge", p0, [(-(3.0 + rng.uniform()) * p0 + 0.05 * rng.standard_normal(S)).astype(F32) for _ in range(steps)], 5e-2)
    np.savez(os.path.join(HERE, "riemann.npz"), **out)


# ------------------------------------------------------------------------------ area / DTW errors by the reference's own functions
def gen_area_dtw(N=300, S=100):
    """`_area_error` (:780-812) and `_dtw_error` (:815-863) of the reference run as they stand, with
    `scipy.integrate.trapz` aliased to its current name and `pyts.metrics.dtw` served by refharness's independent DTW;
    then `reconstruction_errors` / `score_anomalies` (:407-576, :866-962) for both error types."""
    sc = load_npz("score.npz")
    y, y_hat, critic = sc["y"], sc["y_hat"], sc["critic"]
    out = {}
    rng = np.random.default_rng(77)
    # the two error functions on their own: a general series, one shorter than the DTW length, one of exactly 11/12 samples
    for tag, n in (("a", 257), ("b", 7), ("c", 11), ("d", 12), ("e", 30)):
        t = np.sin(np.arange(n) / 9.0) + 0.1 * rng.standard_normal(n)
        p = t + 0.2 * rng.standard_normal(n)
        out[f"ser_{tag}_true"], out[f"ser_{tag}_pred"] = t, p
        out[f"area_{tag}"] = np.asarray(ref_adu._area_error(t, p, 10), dtype=np.float64)
        out[f"dtw_{tag}"] = np.asarray(ref_adu._dtw_error(t, p, 10), dtype=np.float64)
    out["area_sw6"] = np.asarray(ref_adu._area_error(out["ser_a_true"], out["ser_a_pred"], 6), dtype=np.float64)
    out["dtw_sw6"] = np.asarray(ref_adu._dtw_error(out["ser_a_true"], out["ser_a_pred"], 6), dtype=np.float64)
    out["dtw_sw7"] = np.asarray(ref_adu._dtw_error(out["ser_a_true"], out["ser_a_pred"], 7), dtype=np.float64)
    w = int(N * 0.01)
    for kind in ("area", "dtw"):
        raw, _ = ref_adu.reconstruction_errors(y, y_hat, 1, 10, w, False, kind)
        sm, _ = ref_adu.reconstruction_errors(y, y_hat, 1, 10, w, True, kind)
        out[f"rec_{kind}_raw"] = np.asarray(raw, dtype=np.float64)
        out[f"rec_{kind}_smooth"] = np.asarray(sm, dtype=np.float64)
        for comb in ("mult", "sum", "rec"):
            fs, _, _, _ = ref_adu.score_anomalies(y, y_hat, critic, None, rec_error_type=kind, comb=comb)
            out[f"eucl_{kind}_{comb}"] = np.asarray(fs, dtype=np.float64)
    # the DTW stand-in itself on a few pairs, so that tests can hold the oracle's recurrence against it
    xs = rng.standard_normal((6, 11)); ys = rng.standard_normal((6, 11))
    from pyts.metrics import dtw as stub_dtw
    out.update(dtw_pairs_x=xs, dtw_pairs_y=ys, dtw_pairs_out=np.array([stub_dtw(a, b) for a, b in zip(xs, ys)]))
    np.savez(os.path.join(HERE, "score_area_dtw.npz"), **out)


def load_npz(name):
    return dict(np.load(os.path.join(HERE, name), allow_pickle=False))


# ------------------------------------------------------------------------------ interval extraction + metrics
def gen_intervals():
    """find_anomalies (:1363-1472) and its helpers, contextual_confusion_matrix (:606-655, weighted=False)."""
    import pandas as pd
    rng = np.random.default_rng(11)
    out = {}
    cases = []

    def series(n, spikes, width, seed, plateau=False):
        r = np.random.default_rng(seed)
        e = 1.0 + 0.1 * np.abs(r.standard_normal(n))
        for k in range(spikes):
            c = int(r.integers(0, n))
            w = int(r.integers(1, width + 1))
            e[c: c + w] += r.uniform(0.8, 3.0) if not plateau else 2.0
        return e

    specs = [  # name, errors, kwargs
        ("uni", series(2015, 3, 30, 1), dict(window_size_portion=0.33, window_step_size_portion=0.1, fixed_threshold=True)),
        ("uni_edge", series(700, 4, 10, 2), dict(window_size_portion=0.33, window_step_size_portion=0.1, fixed_threshold=True)),
        ("whole", series(900, 2, 40, 3), dict(fixed_threshold=True)),
        ("nopad", series(900, 5, 5, 4), dict(window_size=300, window_step_size=100, fixed_threshold=True, anomaly_padding=0)),
        ("pad5_lower", series(1200, 4, 20, 5), dict(window_size=400, window_step_size=150, fixed_threshold=True, anomaly_padding=5,
                                                  lower_threshold=True, min_percent=0.05)),
        ("plateau", series(1000, 6, 15, 6, plateau=True), dict(window_size_portion=0.5, window_step_size_portion=0.25, fixed_threshold=True,
                                                             anomaly_padding=10)),
        ("flat", np.ones(400), dict(window_size_portion=0.33, window_step_size_portion=0.1, fixed_threshold=True)),
        ("startspike", np.concatenate([np.full(8, 9.0), series(600, 1, 10, 7)]), dict(window_size=200, window_step_size=50,
                                                                                      fixed_threshold=True, anomaly_padding=3)),
        ("dynamic", series(600, 3, 12, 8), dict(window_size=300, window_step_size=150, fixed_threshold=False, anomaly_padding=10)),
    ]
    for name, e, kw in specs:
        idx = np.arange(1000, 1000 + 7 * len(e), 7, dtype=np.int64)        # a non-trivial index (timestamps)
        raised = ""
        try:
            res = ref_adu.find_anomalies(e.copy(), idx, **kw)
        except ZeroDivisionError as ex:       # np.average over zero-length sequences (:1302): the reference raises
            res, raised = [], type(ex).__name__
        out[f"fa_{name}_errors"] = e
        out[f"fa_{name}_index"] = idx
        out[f"fa_{name}_out"] = np.asarray(res, dtype=np.float64).reshape(-1, 3) if len(res) else np.zeros((0, 3))
        cases.append(dict(name=name, kwargs=kw, raises=raised))
    out["fa_cases"] = np.array(json.dumps(cases))
    # helpers on one window
    e = series(500, 3, 15, 21)
    thr = ref_adu._fixed_threshold(e)
    seqs, max_below = ref_adu._find_sequences(e, thr, 7)
    me = ref_adu._get_max_errors(e, seqs, max_below)
    pr = ref_adu._prune_anomalies(me, 0.1)
    out.update(h_errors=e, h_threshold=np.float64(thr), h_sequences=np.asarray(seqs, dtype=np.int64), h_max_below=np.float64(max_below),
               h_max_errors=me[["start", "stop", "max_error"]].values.astype(np.float64), h_pruned=np.asarray(pr, dtype=np.float64),
               h_scores=np.asarray(ref_adu._compute_scores(pr, e, thr, 40), dtype=np.float64))
    out["h_dyn_threshold"] = np.float64(ref_adu._find_threshold(e, (0, 10)))
    out["h_zcost"] = np.array([ref_adu.z_cost(z, e, e.mean(), e.std()) for z in (0.5, 2.0, 4.0, 50.0)])
    merged_in = [[10, 20, 1.0], [21, 30, 3.0], [5, 8, 0.5], [100, 140, 2.0], [120, 130, 4.0], [141, 141, 7.0], [300, 310, 1.5]]
    out["merge_in"] = np.asarray(merged_in, dtype=np.float64)
    out["merge_out"] = np.asarray(ref_adu._merge_sequences([list(m) for m in merged_in]), dtype=np.float64)
    # confusion matrix, overlap-segment form
    cm_cases = [
        ([(10, 20), (50, 60), (100, 110)], [(15, 18), (19, 55), (200, 210), (300, 305)]),
        ([(10, 20)], []),
        ([], [(1, 2), (5, 9)]),
        ([(0, 5), (6, 9)], [(5, 6)]),
        ([(100, 200)], [(90, 100), (200, 210), (150, 160)]),
    ]
    for k, (ex, ob) in enumerate(cm_cases):
        out[f"cm_{k}_expected"] = np.asarray(ex, dtype=np.int64).reshape(-1, 2)
        out[f"cm_{k}_observed"] = np.asarray(ob, dtype=np.int64).reshape(-1, 2)
        r = ref_adu.contextual_confusion_matrix([tuple(x) for x in ex], [tuple(x) for x in ob], weighted=False)
        out[f"cm_{k}_out"] = np.array([-1 if v is None else v for v in r], dtype=np.int64)
    # DataFrame inputs, as univariate_anomaly_detection passes them (:101-108)
    kn = pd.DataFrame({"start": [1400, 5000], "end": [1900, 5600]})
    pa = pd.DataFrame(out["fa_uni_out"], columns=["start", "end", "score"])
    r = ref_adu.contextual_confusion_matrix(kn, pa, data=pd.DataFrame({"timestamp": out["fa_uni_index"]}), weighted=False)
    out["cm_df_known"] = kn.values.astype(np.int64)
    out["cm_df_out"] = np.array([-1 if v is None else v for v in r], dtype=np.int64)
    # casas_anomalies (:279-298): label runs -> ground-truth intervals, with its cut-the-last-point behaviour
    r2 = np.random.default_rng(3)
    for k in range(6):
        nb, b = int(r2.integers(1, 6)), int(r2.integers(4, 40))
        y = (r2.random((nb, b, 1)) < r2.uniform(0.05, 0.6)).astype(np.float32)
        if k % 2 == 0:
            y[-1, -3:] = 1          # a run still open at the end
        if k % 3 == 0:
            y[0, :2] = 1            # a run from the first point
        if k == 5:
            y[0, 0], y[0, 1] = 1, 0  # a one-point run at index 0: the reference's end index wraps
        n = int(r2.integers(max(1, nb * b - 5), nb * b + 1))
        x = 1000.0 + 3.0 * np.arange(n)
        out[f"casas_{k}_y"], out[f"casas_{k}_x"] = y, x
        out[f"casas_{k}_out"] = ref_adu.casas_anomalies(torch.from_numpy(y), x).values.astype(np.float64).reshape(-1, 2)
    np.savez(os.path.join(HERE, "intervals.npz"), **out)


# ------------------------------------------------------------------------------ data pipeline
def synth_signal_csv(path, n, seed, step=300, gaps=True, yahoo=False):
    """A NAB-style CSV (timestamp,value): irregular sampling, missing stretches, a NaN or two."""
    import pandas as pd
    r = np.random.default_rng(seed)
    if yahoo:
        ts = np.arange(1, n + 1)
        val = 0.01 * ts + np.sin(ts / 24.0) + 0.1 * r.standard_normal(n)
        anom = np.zeros(n, dtype=np.int64)
        anom[n // 3: n // 3 + 7] = 1
        anom[n - 40: n - 35] = 1
        val[anom == 1] += 3
        pd.DataFrame({"timestamp": ts, "value": val, "is_anomaly": anom}).to_csv(path, index=False)
        return
    ts = 1_400_000_000 + step * np.arange(n) + r.integers(0, step // 3, n)
    val = 50 + 20 * np.sin(np.arange(n) / 40.0) + r.standard_normal(n)
    keep = np.ones(n, dtype=bool)
    if gaps:
        keep[n // 4: n // 4 + 30] = False            # a missing stretch: empty buckets -> NaN -> imputed
        keep[n // 2: n // 2 + 3] = False
        val[n // 5] = np.nan
    order = r.permutation(int(keep.sum()))             # unsorted rows: the reference sorts by time stamp
    pd.DataFrame({"timestamp": ts[keep][order], "value": val[keep][order]}).to_csv(path, index=False)


def gen_dataloader():
    """utils/dataloader.py:61-232 on synthetic CSVs written to a temporary directory."""
    import tempfile
    import utils.dataloader as ref_dl
    out, cases = {}, []
    with tempfile.TemporaryDirectory() as tmp:
        specs = [("nab600", dict(n=1500, seed=1, step=300), dict(interval=600, windows_size=100)),
                 ("nab1800", dict(n=2000, seed=2, step=300), dict(interval=1800, windows_size=50)),
                 ("dense", dict(n=900, seed=3, step=300, gaps=False), dict(interval=300, windows_size=100)),
                 ("coarse", dict(n=4000, seed=4, step=60), dict(interval=3600, windows_size=30)),
                 ("yahoo", dict(n=700, seed=5, yahoo=True), dict(interval=1, windows_size=100, yahoo=True))]
        for name, ckw, dkw in specs:
            path = os.path.join(tmp, f"{name}.csv")
            synth_signal_csv(path, **ckw)
            ds = ref_dl.SignalDataset(path, test=True, **dkw)
            out[f"dl_{name}_csv"] = np.array(open(path).read())
            out[f"dl_{name}_index"] = np.asarray(ds.index)
            out[f"dl_{name}_series"] = np.concatenate([ds.X[0, :, 0], ds.X[1:, -1, 0]])      # the scaled series the windows slide over
            out[f"dl_{name}_Xshape"] = np.asarray(ds.X.shape)
            out[f"dl_{name}_Xrows"] = ds.X[::max(1, len(ds.X) // 7)]
            out[f"dl_{name}_y"] = ds.y
            out[f"dl_{name}_X_index"] = ds.X_index
            out[f"dl_{name}_y_index"] = ds.y_index
            if dkw.get("yahoo"):
                import pandas as pd
                out[f"dl_{name}_known"] = pd.read_csv(path[:-4] + "_known_anomalies.csv")[["start", "end"]].values
            cases.append(dict(name=name, dataset=dkw))
        out["dl_cases"] = np.array(json.dumps(cases))
        # the two building blocks on their own, with the arguments the reference never exercises
        Xa = np.column_stack([np.arange(0, 400, 7), np.random.default_rng(0).standard_normal(58), np.arange(58) % 5.0])
        ds0 = ref_dl.SignalDataset.__new__(ref_dl.SignalDataset)
        v, i = ds0.time_segments_aggregate(Xa, 50, 0, method=["mean", "max"])
        out.update(tsa_in=Xa, tsa_values=v, tsa_index=i)
        Xr = np.random.default_rng(1).standard_normal((60, 2))
        Xr[17, 0] = np.nan
        Xr[41, 1] = np.nan
        for tag, kw in (("plain", dict(window_size=8, target_size=2, step_size=3, target_column=1, offset=1)),
                        ("drop", dict(window_size=8, target_size=1, step_size=2, target_column=0, drop=float("nan"), drop_windows=True))):
            a, b, c, d = ds0.rolling_window_sequences(Xr, np.arange(100, 160), **kw)
            out.update({f"rws_{tag}_X": a, f"rws_{tag}_y": b, f"rws_{tag}_Xi": c, f"rws_{tag}_yi": d})
        out["rws_in"] = Xr
    np.savez_compressed(os.path.join(HERE, "dataloader.npz"), **out)


def gen_multivariate():
    """utils/dataloader_multivariate.py:16-121 on synthetic tensors / CSVs written to a temporary directory (the reference
    reads the SWaT / WADI files relative to the working directory, so it is run from there)."""
    import tempfile
    import pandas as pd
    import utils.dataloader_multivariate as ref_mv
    rng = np.random.default_rng(11)
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            # CASAS / ELINUS / eHealth: (n, 5, 30) sequences -> (n, 150), MinMax per column; one constant column
            seq = rng.standard_normal((97, 5, 30)).astype(np.float32) * 3 + 1
            seq[:, 2, 7] = 0.25
            gt = (rng.random(97) < 0.1).astype(np.int64)
            torch.save(torch.from_numpy(seq), "seq.pt"); torch.save(torch.from_numpy(gt), "gt.pt")
            for test in (False, True):
                ds = ref_mv.MultivariateDataset(seq_path="seq.pt", gt_path="gt.pt", test=test, dataset="CASAS")
                out[f"mv_casas_X_{int(test)}"] = np.asarray(ds.X)
            out.update(mv_casas_seq=seq, mv_casas_gt=gt)
            # new_CASAS: directory with x_train / y_train / x_test / y_test
            os.makedirs("nc")
            for part, n in (("train", 64), ("test", 41)):
                xx = rng.uniform(-5, 9, (n, 150)).astype(np.float32)
                yy = (rng.random(n) < 0.2).astype(np.int64)
                torch.save(torch.from_numpy(xx), f"nc/x_{part}"); torch.save(torch.from_numpy(yy), f"nc/y_{part}")
                ds = ref_mv.MultivariateDataset(seq_path="nc", gt_path="nc", test=(part == "test"), dataset="new_CASAS")
                out.update({f"mv_nc_x_{part}": xx, f"mv_nc_y_{part}": yy, f"mv_nc_X_{part}": np.asarray(ds.X)})
            # CASAS_: (a, b, 5) rows, first 4500 dropped, train = rows before (first anomaly - 1000), test = +-1000 around them
            X = rng.standard_normal((80, 100, 5)).astype(np.float32)
            y = np.zeros((80, 100, 1), dtype=np.int64)
            y.reshape(-1)[6520:6621] = 1
            torch.save(torch.from_numpy(X), "cX.pt"); torch.save(torch.from_numpy(y), "cy.pt")
            for test in (False, True):
                ds = ref_mv.MultivariateDataset(seq_path="cX.pt", gt_path="cy.pt", test=test, dataset="CASAS_")
                out[f"mv_casas__X_{int(test)}"] = np.asarray(ds.X)
                out[f"mv_casas__y_{int(test)}"] = np.asarray(ds.y)
            out.update(mv_casas__seq=X, mv_casas__gt=y)
            # SWaT / WADI CSVs with gaps
            os.makedirs("data/SWAT"); os.makedirs("data/WADI_downsampled")
            def frame(n, k, seed):
                r = np.random.default_rng(seed)
                a = r.standard_normal((n, k)) * r.uniform(0.5, 20, k) + r.uniform(-3, 3, k)
                a[r.random((n, k)) < 0.03] = np.nan
                a[:, 1] = 4.0
                return pd.DataFrame(a, columns=[f"f{i}" for i in range(k)])
            tr, te = frame(120, 6, 1), frame(90, 6, 2)
            tr.insert(0, "Timestamp", np.arange(120)); tr["Normal/Attack"] = "Normal"
            te.insert(0, "Timestamp", np.arange(90)); te["Normal/Attack"] = "Attack"; te["label"] = 1
            tr.to_csv("data/SWAT/SWaT_train_mine.csv"); te.to_csv("data/SWAT/SWaT_test_mine.csv")
            wtr, wte = frame(110, 7, 3), frame(70, 7, 4)
            wte.insert(0, "Time", np.arange(70)); wte["label"] = 0
            wtr.to_csv("data/WADI_downsampled/WADI_train.csv", index=False); wte.to_csv("data/WADI_downsampled/WADI_test_mine.csv", index=False)
            for name in ("SWAT", "WADI"):
                for test in (False, True):
                    ds = ref_mv.MultivariateDataset(test=test, dataset=name)
                    out[f"mv_{name}_X_{int(test)}"] = np.asarray(ds.X)
            for rel in ("data/SWAT/SWaT_train_mine.csv", "data/SWAT/SWaT_test_mine.csv", "data/WADI_downsampled/WADI_train.csv",
                        "data/WADI_downsampled/WADI_test_mine.csv"):
                out["mv_csv_" + os.path.basename(rel)[:-4]] = np.array(open(rel).read())
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "multivariate.npz"), **out)


if __name__ == "__main__":
    import pandas
    import scipy
    if len(sys.argv) > 1 and sys.argv[1] == "riemann":
        gen_riemann()
        print("riemann.npz written")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "area_dtw":
        gen_area_dtw()
        print("score_area_dtw.npz written")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "multivariate":
        gen_multivariate()
        print("multivariate.npz written")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dataloader":
        gen_dataloader()
        print("dataloader.npz written")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "intervals":
        gen_intervals()
        print("intervals.npz written")
        sys.exit(0)
    gen_forward(100, 64, "S100_B64")
    gen_forward(150, 256, "S150_B256")
    gen_ops()
    gen_iters(100, 64, True, "hyper_S100")
    gen_iters(100, 64, False, "eucl_S100")
    gen_scoring()
    gen_riemann()
    gen_area_dtw()
    gen_intervals()
    gen_dataloader()
    with open(os.path.join(HERE, "versions.json"), "w") as f:
        json.dump(dict(torch=torch.__version__, numpy=np.__version__, scipy=scipy.__version__,
                       pandas=pandas.__version__, python=sys.version.split()[0],
                       reference="aleflabo/HypAD @ /root/reference (v1)",
                       note="RiemannianAdam steps (iters_hyper: w1/wN of dec.* and enc.*) come from oracle.radam "
                            "(geoopt 0.5.0 is not vendored): UNPINNED"), f, indent=1)
    print("fixtures written to", HERE)
