"""GPU parity against the round-2 fixture groups (all produced by the reference's own code, tests/golden/gen_fixtures.py):
riemann.npz (Riemannian primitives of math_.py composed into geoopt's RiemannianAdam step) and score_area_dtw.npz
(`_area_error`, `_dtw_error`, `score_anomalies` for both error types).  Everything goes through the C ABI."""
import numpy as np
import pytest
import torch

from helpers import load, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda")


@pytest.mark.parametrize("tag", ["init", "mid", "edge"])
def test_radam_ball_step_follows_reference_primitives(dev, tag):
    """hypad_radam_step (ball branch) vs the 23-step trajectories assembled from math_.py:340-352, 419-430, 656-676,
    1738-1746, 1843-1845 in geoopt 0.5.0's order; `stabilize` fires at steps 10 and 20; `edge` hits the projection."""
    from hypad_amd import optim as ho
    from hypad_amd.hyperspace.hyrnn_nets import ManifoldParameter, PoincareBall
    fx = load("riemann.npz")
    p = ManifoldParameter(torch.from_numpy(fx[f"traj_{tag}_p0"].copy()).cuda(), manifold=PoincareBall())
    opt = ho.RiemannianAdam([p], lr=float(fx[f"traj_{tag}_lr"]), weight_decay=1e-5, stabilize=10)
    for t, g in enumerate(fx[f"traj_{tag}_grads"]):
        p.grad = torch.from_numpy(g.copy()).cuda()
        opt.step()
        st = opt.state[p]
        ref_p, ref_m, ref_v = fx[f"traj_{tag}_p"][t], fx[f"traj_{tag}_m"][t], fx[f"traj_{tag}_v"][t]
        assert maxdiff(p.detach().cpu(), ref_p) < 2e-6, (tag, t, maxdiff(p.detach().cpu(), ref_p))
        assert maxdiff(st["exp_avg"].cpu(), ref_m) < 1e-5 * max(1.0, np.abs(ref_m).max()), (tag, t)
        assert maxdiff(st["exp_avg_sq"].cpu(), ref_v) < 1e-5 * max(1.0, np.abs(ref_v).max()), (tag, t)


def test_area_and_dtw_kernels_match_the_reference_functions(dev):
    """hypad_area_error / hypad_dtw_error / the Euclidean scoring pipeline vs the reference's `_area_error` (:780-812),
    `_dtw_error` (:815-863) and `score_anomalies(rec_error_type=area|dtw)` (:407-576)."""
    from hypad_amd.utils import anomaly_detection_utils as adu
    fx = load("score_area_dtw.npz")
    for tag in "abcde":
        t, p = fx[f"ser_{tag}_true"], fx[f"ser_{tag}_pred"]
        # the kernels take the un-rolled prediction as float32 (it is a median of float32 reconstructions): feed the
        # float32-rounded series to both sides' inputs is not possible for a fixed fixture, so compare at 1e-6
        got_a = adu._area_error(t, p, 10).cpu().numpy()
        assert np.allclose(got_a, fx[f"area_{tag}"], rtol=0, atol=2e-6, equal_nan=True), tag
        got_d = adu._dtw_error(t, p, 10).cpu().numpy()
        assert got_d.shape == fx[f"dtw_{tag}"].shape and maxdiff(got_d, fx[f"dtw_{tag}"]) < 2e-6, tag
        assert np.array_equal(got_d == 0, fx[f"dtw_{tag}"] == 0), tag            # zero framing: exactly the same positions
    t, p = fx["ser_a_true"], fx["ser_a_pred"]
    assert np.allclose(adu._area_error(t, p, 6).cpu().numpy(), fx["area_sw6"], rtol=0, atol=2e-6, equal_nan=True)
    assert maxdiff(adu._dtw_error(t, p, 6).cpu().numpy(), fx["dtw_sw6"]) < 2e-6
    assert maxdiff(adu._dtw_error(t, p, 7).cpu().numpy(), fx["dtw_sw7"]) < 2e-6
    sc = load("score.npz")
    y, y_hat, critic = sc["y"], sc["y_hat"], sc["critic"]
    w = int(len(y) * 0.01)
    for kind in ("area", "dtw"):
        raw, _ = adu.reconstruction_errors(y, y_hat, 1, 10, w, False, kind, with_summary=False)
        sm, _ = adu.reconstruction_errors(y, y_hat, 1, 10, w, True, kind, with_summary=False)
        assert np.allclose(raw, fx[f"rec_{kind}_raw"], rtol=0, atol=1e-9, equal_nan=True), kind
        assert np.allclose(sm, fx[f"rec_{kind}_smooth"], rtol=0, atol=1e-9, equal_nan=True), kind
        for comb in ("mult", "sum", "rec"):
            got, _, _, _ = adu.score_anomalies(y, y_hat, critic, None, rec_error_type=kind, comb=comb)
            assert np.allclose(got, fx[f"eucl_{kind}_{comb}"], rtol=0, atol=1e-6, equal_nan=True), (kind, comb)
