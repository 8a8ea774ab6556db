"""Pin the CPU oracle against the golden vectors produced by the reference itself
(tests/golden/gen_fixtures.py).  CPU only; no HIP involved."""
import numpy as np
import pytest
import torch

from helpers import load, maxdiff, oracle_models, params_ns, sub_state
from oracle import gmath, scoring, train_iters
from oracle.radam import RiemannianAdam

torch.set_num_threads(1)


@pytest.mark.parametrize("tag,S,B", [("S100_B64", 100, 64), ("S150_B256", 150, 256)])
def test_networks_forward(tag, S, B):
    fx = load(f"fwd_{tag}.npz")
    enc, dec, cx, cz = oracle_models(fx, S, True)
    for m in (enc, dec, cx, cz):
        m.eval()
    x, z = torch.from_numpy(fx["x"]), torch.from_numpy(fx["z"]).view(1, B, 20)
    with torch.no_grad():
        hyper, eucl = dec(z)
        assert maxdiff(enc(x), fx["enc_x"]) < 1e-6
        assert maxdiff(hyper, fx["dec_hyper"]) < 1e-6
        assert maxdiff(eucl, fx["dec_eucl"]) < 1e-6
        assert maxdiff(dec.hyperbolic_linear(x.view(-1, S).float()), fx["head_x"]) < 1e-6
        assert maxdiff(cx(x), fx["cx_x"]) < 1e-6
        assert maxdiff(cz(z), fx["cz_z"]) < 1e-6
        h2, e2 = dec(enc(x.float()))
        assert maxdiff(h2, fx["s0_hyper"]) < 1e-6 and maxdiff(e2, fx["s0_eucl"]) < 1e-6
    _, dec_e, _, _ = oracle_models(fx, S, False)
    dec_e.eval()
    with torch.no_grad():
        assert maxdiff(dec_e(z), fx["dec_e_out"]) < 1e-6


def _check_op(fx, name, fn, *keys, tol=1e-6, gtol=None):
    ins = [torch.from_numpy(fx[k].copy()).requires_grad_(True) for k in keys]
    out = fn(*ins)
    assert maxdiff(out.detach(), fx[f"{name}_out"]) <= tol, name
    gs = torch.autograd.grad(out, ins, torch.from_numpy(fx[f"{name}_gout"]), allow_unused=True)
    for i, g in enumerate(gs):
        ref = fx[f"{name}_gin{i}"]
        scale = max(1.0, float(np.max(np.abs(ref))))
        assert maxdiff(g, ref) <= (gtol or tol) * scale, (name, i)


def test_hyperbolic_ops_and_gradients():
    fx = load("ops.npz")
    _check_op(fx, "expmap0", gmath.expmap0, "u")
    _check_op(fx, "logmap0", gmath.logmap0, "ball")
    _check_op(fx, "mobius_add", gmath.mobius_add, "ball", "y2")
    _check_op(fx, "mobius_add_bias", lambda a, b: gmath.mobius_add(a, b.unsqueeze(0).expand_as(a)), "ball", "bias_big")
    _check_op(fx, "project", gmath.project, "u")
    fx["u_half"] = fx["u"][:120] * 0.5
    fx["W_small"] = fx["W"] * 0.01
    _check_op(fx, "mobius_linear", gmath.mobius_linear, "u_half", "W", "bias_big")
    _check_op(fx, "mobius_linear_small", gmath.mobius_linear, "u_half", "W_small", "bias")
    inside = fx["ball"][:80]
    fx["rd_a"], fx["rd_b"] = inside, np.roll(inside, 3, axis=0) * 0.9
    _check_op(fx, "rowdist", gmath.rowwise_poincare_distance, "rd_a", "rd_b", gtol=1e-5)
    fx["pa"] = np.concatenate([inside[:30], inside[:2], np.zeros((2, 100), np.float32)])
    fx["pb"] = np.concatenate([inside[40:70] * 0.8, inside[:3]])
    _check_op(fx, "pairdist", gmath.pairwise_poincare_distance, "pa", "pb", gtol=1e-5)
    out = gmath.rowwise_poincare_distance(torch.from_numpy(inside), torch.from_numpy(inside.copy()))
    assert maxdiff(out, fx["rowdist_same_out"]) < 1e-6


def test_manifold_identities():
    """properties the reference never tests but geoopt documents."""
    g = torch.Generator().manual_seed(0)
    u = torch.randn(64, 100, generator=g) * 0.05
    assert maxdiff(gmath.logmap0(gmath.expmap0(u)), u) < 1e-6
    x, y = gmath.expmap0(u), gmath.expmap0(torch.randn(64, 100, generator=g) * 0.03)
    assert maxdiff(gmath.mobius_add(-x, gmath.mobius_add(x, y)), y) < 1e-6          # left cancellation, math_.py:511-515
    big = torch.randn(8, 100, generator=g)
    assert float(gmath.project(big).norm(dim=-1).max()) <= 1 - 4e-3 + 1e-6          # math_.py:343-352
    d1, d2 = gmath.rowwise_poincare_distance(x, y), gmath.rowwise_poincare_distance(y, x)
    assert maxdiff(d1, d2) < 1e-6


def _run_iters(tag, hyperbolic):
    fx = load(f"iters_{tag}.npz")
    S, B = 100, 64
    enc, dec, cx, cz = oracle_models(fx, S, hyperbolic, wkey="w0")
    for m in (enc, dec, cx, cz):
        m.eval()
    P = params_ns(B, S, hyperbolic)
    ocx, ocz, odec = train_iters.make_optimizers(enc, dec, cx, cz, P)
    samples = [torch.from_numpy(s) for s in fx["samples"]]
    steps = len(samples)
    train_iters.set_trainable((enc, dec), False)
    train_iters.set_trainable((cx, cz), True)
    got = dict(cx=[], cz=[], dec=[], hyp=[], mse=[])
    for i in range(steps):
        l = train_iters.critic_x_iteration(samples[i], dec, cx, ocx, P, z=fx["z_cx"][i], alpha=fx["a_cx"][i])
        got["cx"].append(float(l))
        if i == 0:
            assert str(l.dtype) == str(fx["cx_loss_dtype"])
            for k, p in cx.named_parameters():
                assert maxdiff(p.grad, fx[f"g1.cx_iter.cx.{k}"]) < 2e-6, k
            for k, v in cx.state_dict().items():
                assert maxdiff(v, fx[f"w1.cx.{k}"]) < 1e-6, k
        l = train_iters.critic_z_iteration(samples[i], enc, cz, ocz, P, z=fx["z_cz"][i], alpha=fx["a_cz"][i])
        got["cz"].append(float(l))
        if i == 0:
            for k, p in cz.named_parameters():
                assert maxdiff(p.grad, fx[f"g1.cz_iter.cz.{k}"]) < 2e-6, k
    assert maxdiff(got["cx"], fx["loss_cx"]) < 1e-5 and maxdiff(got["cz"], fx["loss_cz"]) < 1e-5
    for k, v in cx.state_dict().items():
        assert maxdiff(v, fx[f"wN.cx.{k}"]) < 2e-5, k
    for k, v in cz.state_dict().items():
        assert maxdiff(v, fx[f"wN.cz.{k}"]) < 2e-5, k
    train_iters.set_trainable((enc, dec), True)
    train_iters.set_trainable((cx, cz), False)
    for i in range(steps):
        l, h, m = train_iters.decoder_iteration(samples[i], enc, dec, cx, cz, odec, P, z=fx["z_dec"][i])
        got["dec"].append(float(l)); got["hyp"].append(float(h)); got["mse"].append(float(m))
        if i == 0:
            for name, mod in (("dec", dec), ("enc", enc)):
                for k, p in mod.named_parameters():
                    ref = fx[f"g1.dec_iter.{name}.{k}"]
                    assert maxdiff(p.grad, ref) < 2e-6 * max(1.0, float(np.abs(ref).max())), (name, k)
                for k, v in mod.state_dict().items():
                    assert maxdiff(v, fx[f"w1.{name}.{k}"]) < 1e-6, (name, k)
    assert maxdiff(got["dec"], fx["loss_dec"]) < 2e-5
    assert maxdiff(got["hyp"], fx["loss_hyper"]) < 2e-5
    assert maxdiff(got["mse"], fx["loss_mse"]) < 2e-5
    for name, mod in (("dec", dec), ("enc", enc)):
        for k, v in mod.state_dict().items():
            assert maxdiff(v, fx[f"wN.{name}.{k}"]) < 5e-5, (name, k)


def test_training_iterations_hyperbolic():
    _run_iters("hyper_S100", True)


def test_training_iterations_euclidean():
    _run_iters("eucl_S100", False)


def test_riemannian_adam_euclidean_branch_equals_torch_adam_l2():
    """The only pin available for geoopt's optimizer (oracle/radam.py header)."""
    torch.manual_seed(0)
    w0 = torch.randn(37, 11)
    a, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    oa = RiemannianAdam([a], lr=5e-4, weight_decay=1e-5, stabilize=10)
    ob = torch.optim.Adam([b], lr=5e-4, weight_decay=1e-5)
    for _ in range(25):
        g = torch.randn(37, 11)
        a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    assert maxdiff(a.detach(), b.detach()) < 1e-6


def test_riemannian_adam_ball_branch_stays_on_ball_and_descends():
    from oracle.tadgan import BallParameter
    torch.manual_seed(1)
    target = gmath.expmap0(torch.randn(1, 100) * 0.05)
    p = BallParameter(gmath.expmap0(torch.randn(100) * 0.02))
    opt = RiemannianAdam([p], lr=5e-3, weight_decay=1e-5, stabilize=10)
    first = None
    for _ in range(200):
        opt.zero_grad()
        d = gmath.rowwise_poincare_distance(p.unsqueeze(0), target).sum()
        first = float(d) if first is None else first
        d.backward()
        opt.step()
        assert float(p.norm()) <= 1 - 4e-3 + 1e-6
    assert float(d) < 0.2 * first


def test_scoring_pins():
    fx = load("score.npz")
    y, y_hat, critic = fx["y"], fx["y_hat"], fx["critic"]
    n = len(y)
    w = int(n * 0.01)
    assert maxdiff(scoring.unroll_true(y), fx["true_unrolled"]) == 0
    err, summ = scoring.reconstruction_errors(y, y_hat, 10, w, True, "point")
    assert np.allclose(err, fx["point_err"], rtol=0, atol=1e-12, equal_nan=True)
    assert maxdiff(summ, fx["predictions_vs"]) < 1e-12
    raw, _ = scoring.reconstruction_errors(y, y_hat, 10, w, False, "point", with_summary=False)
    assert maxdiff(raw, fx["point_err_raw"]) < 1e-12
    assert maxdiff(scoring.zscore_clip(err), fx["point_z"]) < 1e-12
    cs = scoring.final_critic_scores(critic, n, y.shape[1])
    assert maxdiff(cs, fx["critic_scores"]) < 1e-10
    assert np.allclose(scoring.compute_critic_score(critic, 7), fx["critic_score_direct"], atol=1e-12, equal_nan=True)
    fs, crit, rec = scoring.hyperbolic_scores(fx["ball_recons"], fx["ball_real"], critic, "mult")
    assert maxdiff(rec, fx["hyper_rec"]) < 1e-6
    for comb in ("sum", "mult", "uncertainty", "critic", "critic_uncertainty", "sum_uncertainty", "rec", "rec_uncertainty"):
        got = scoring.combine_scores(comb, crit, rec, fx["ball_recons"])
        assert maxdiff(got, fx[f"comb_{comb}"]) < 1e-6, comb
    for comb in ("mult", "sum", "rec", "critic"):
        got, _, _ = scoring.score_anomalies(y, y_hat, critic, "point", comb)
        assert maxdiff(got, fx[f"eucl_{comb}"]) < 1e-10, comb


def test_dtw_and_area_known_answers():
    """pyts is absent (UNPINNED): brute force + hand-computed cases."""
    assert scoring.dtw_classic([0, 0, 0], [0, 0, 0]) == 0
    assert abs(scoring.dtw_classic([0, 1, 2], [0, 1, 2])) < 1e-15
    # hand: x=[0,2], y=[1,1] -> C=[[1,1],[1,1]]; D=[[1,2],[2,2]] -> sqrt(2)
    assert abs(scoring.dtw_classic([0, 2], [1, 1]) - 2 ** 0.5) < 1e-15
    # warping absorbs a repeated sample: [0,1,1,2] vs [0,1,2,2]
    assert scoring.dtw_classic([0, 1, 1, 2], [0, 1, 2, 2]) == 0

    def brute(x, y):
        import itertools
        n, m = len(x), len(y)
        best = [np.inf]

        def walk(i, j, acc):
            acc += (x[i] - y[j]) ** 2
            if i == n - 1 and j == m - 1:
                best[0] = min(best[0], acc); return
            for di, dj in ((1, 0), (0, 1), (1, 1)):
                if i + di < n and j + dj < m:
                    walk(i + di, j + dj, acc)
        walk(0, 0, 0.0)
        return best[0] ** 0.5

    rng = np.random.default_rng(0)
    for _ in range(5):
        a, b = rng.standard_normal(6), rng.standard_normal(6)
        assert abs(scoring.dtw_classic(a, b) - brute(a, b)) < 1e-12
    t = rng.standard_normal(60)
    p = t + 0.1 * rng.standard_normal(60)
    e = scoring.dtw_error(t, p)
    assert len(e) == 60 and np.all(e[:5] == 0) and np.all(e[-6:] == 0) and np.all(e[5:-6] >= 0)
    a = scoring.area_error(t, p)
    # centred window of 10 at i covers [i-5, i+4]; trapezoid rule with unit spacing
    i = 20
    manual = abs(np.trapezoid(t[i - 5:i + 5]) - np.trapezoid(p[i - 5:i + 5]))
    assert abs(a[i] - manual) < 1e-12


# ---------------------------------------------------------------------------------------------- round 2 pins
def test_riemannian_primitives_match_the_reference_module():
    """oracle/gmath.py:69-99 against the reference's vendored math_.py (riemann.npz, gen_fixtures.gen_riemann)."""
    fx = load("riemann.npz")
    x, y, u, v = (torch.from_numpy(fx[k]) for k in ("prim_x", "prim_y", "prim_u", "prim_v"))

    def close(got, ref):
        ref = torch.from_numpy(ref)
        return bool(torch.all((got - ref).abs() <= 2e-6 * ref.abs().clamp_min(1.0)))

    assert close(gmath.lambda_x(x, keepdim=True), fx["lambda_x"])
    assert close(gmath.inner(x, u, v, keepdim=True), fx["inner"])
    assert close(gmath.inner(x, u, u, keepdim=True), fx["inner_uu"])
    assert close(gmath.egrad2rgrad(x, u), fx["egrad2rgrad"])
    assert close(gmath.gyration(x, y, u), fx["gyration"])
    assert close(gmath.parallel_transport(x, y, u), fx["parallel_transport"])
    assert close(gmath.project(x), fx["project_x"])


@pytest.mark.parametrize("tag", ["init", "mid", "edge"])
def test_riemannian_adam_ball_trajectory(tag):
    """oracle/radam.py's ball branch vs a 23-step trajectory whose every arithmetic step was a call into the
    reference's math_.py, in geoopt 0.5.0's published order (stabilize at steps 10 and 20 included)."""
    from oracle.tadgan import BallParameter
    fx = load("riemann.npz")
    p = BallParameter(torch.from_numpy(fx[f"traj_{tag}_p0"].copy()))
    opt = RiemannianAdam([p], lr=float(fx[f"traj_{tag}_lr"]), weight_decay=1e-5, stabilize=10)
    for t, g in enumerate(fx[f"traj_{tag}_grads"]):
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        st = opt.state[p]
        assert maxdiff(p.detach(), fx[f"traj_{tag}_p"][t]) < 1e-6, (tag, t)
        assert maxdiff(st["exp_avg"], fx[f"traj_{tag}_m"][t]) < 1e-6 * max(1.0, np.abs(fx[f"traj_{tag}_m"][t]).max()), (tag, t)
        ref_v = fx[f"traj_{tag}_v"][t]
        assert maxdiff(st["exp_avg_sq"], ref_v) < 1e-6 * max(1.0, np.abs(ref_v).max()), (tag, t)
    if tag == "edge":     # the retraction's projection did fire in this trajectory
        assert abs(float(np.linalg.norm(fx["traj_edge_p"][-1])) - (1 - 4e-3)) < 1e-5


def test_area_and_dtw_errors_match_the_reference_functions():
    """oracle/scoring.py area_error / dtw_error / reconstruction_errors / score_anomalies vs the reference's own
    `_area_error` (:780-812), `_dtw_error` (:815-863), `score_anomalies` (:407-576) run by gen_fixtures.gen_area_dtw."""
    fx = load("score_area_dtw.npz")
    for tag in "abcde":
        t, p = fx[f"ser_{tag}_true"], fx[f"ser_{tag}_pred"]
        assert np.allclose(scoring.area_error(t, p, 10), fx[f"area_{tag}"], rtol=0, atol=1e-12, equal_nan=True), tag
        got = scoring.dtw_error(t, p, 10)
        assert got.shape == fx[f"dtw_{tag}"].shape and maxdiff(got, fx[f"dtw_{tag}"]) < 1e-12, tag
    t, p = fx["ser_a_true"], fx["ser_a_pred"]
    assert np.allclose(scoring.area_error(t, p, 6), fx["area_sw6"], rtol=0, atol=1e-12, equal_nan=True)
    assert maxdiff(scoring.dtw_error(t, p, 6), fx["dtw_sw6"]) < 1e-12
    assert maxdiff(scoring.dtw_error(t, p, 7), fx["dtw_sw7"]) < 1e-12
    for a, b, ref in zip(fx["dtw_pairs_x"], fx["dtw_pairs_y"], fx["dtw_pairs_out"]):
        assert abs(scoring.dtw_classic(a, b) - ref) < 1e-12
    sc = load("score.npz")
    y, y_hat, critic = sc["y"], sc["y_hat"], sc["critic"]
    w = int(len(y) * 0.01)
    for kind in ("area", "dtw"):
        raw, _ = scoring.reconstruction_errors(y, y_hat, 10, w, False, kind, with_summary=False)
        sm, _ = scoring.reconstruction_errors(y, y_hat, 10, w, True, kind, with_summary=False)
        assert np.allclose(raw, fx[f"rec_{kind}_raw"], rtol=0, atol=1e-12, equal_nan=True), kind
        assert np.allclose(sm, fx[f"rec_{kind}_smooth"], rtol=0, atol=1e-12, equal_nan=True), kind
        for comb in ("mult", "sum", "rec"):
            got, _, _ = scoring.score_anomalies(y, y_hat, critic, kind, comb)
            assert np.allclose(got, fx[f"eucl_{kind}_{comb}"], rtol=0, atol=1e-9, equal_nan=True), (kind, comb)
