"""Round 3 scoring kernels: the chunked, position-independent centred rolling mean (hypad_rolling_mean: utils/anomaly_detection_utils.py
:953-961, :325-330, with the point-wise error :761-777 fused), the two-level z-score statistics (:523-524, :307-322), the LDS-staged
anti-diagonal un-roll (:918-935)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _series(n, seed, nans=0):
    rng = np.random.default_rng(seed)
    x = np.abs(np.sin(np.arange(n) / 37.0) + 0.3 * rng.standard_normal(n))
    if nans:
        x[rng.integers(0, n, nans)] = np.nan
    return x


@pytest.mark.parametrize("w", [1, 2, 31, 32, 33, 64, 200, 257, 1250, 5000])
def test_rolling_mean_matches_pandas(w):
    from hypad_amd.utils import anomaly_detection_utils as adu
    from oracle import scoring as osc
    for n, nans in ((20_000, 0), (4_099, 25), (300, 0), (17, 0)):
        x = _series(n, w + n, nans)
        got = adu.rolling_mean(x, w).cpu().numpy()
        ref = osc.rolling_mean_centered(x, w)
        assert np.allclose(got, ref, rtol=0, atol=1e-12, equal_nan=True), (w, n)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (w, n)


@pytest.mark.parametrize("w", [7, 41, 200, 1250])
def test_rolling_mean_of_a_slice_has_the_bits_of_the_whole(w):
    """What sharded scoring relies on (parallel.sharded_euclidean_scores): a rank smooths the slice [a, b) of the error series and
    keeps the timesteps whose window lies inside it; passing the slice's position makes those equal, bit for bit, to the
    un-sharded pass -- whatever a and b are."""
    from hypad_amd.utils import anomaly_detection_utils as adu
    x = torch.from_numpy(_series(30_000, w, 40)).cuda()
    whole = adu.rolling_mean(x, w)
    rng = np.random.default_rng(w)
    for _ in range(6):
        a = int(rng.integers(0, 20_000))
        b = int(min(30_000, a + rng.integers(2 * w + 3, 9_000)))
        part = adu.rolling_mean(x[a:b].contiguous(), w, origin=a)
        lo, hi = (0 if a == 0 else w // 2 + 1), (b - a if b == 30_000 else b - a - w // 2 - 1)
        m = ~torch.isnan(whole[a + lo: a + hi])
        assert torch.equal(part[lo:hi][m], whole[a + lo: a + hi][m]) and torch.equal(torch.isnan(part[lo:hi]), ~m), (a, b)
    # the fused point-wise error is the same series, not another rounding of it
    pred = (x.float() + 0.1).contiguous()
    true = torch.nan_to_num(x, nan=0.5)
    fused, plain = adu.rolling_mean(true, w, minus=pred).cpu().numpy(), adu.rolling_mean(adu._point_wise_error(true, pred), w).cpu().numpy()
    assert np.array_equal(fused, plain, equal_nan=True) and not np.isnan(fused[w: -w]).any()


@pytest.mark.parametrize("n", [1, 7, 1000, 125_099, 1_000_003])
def test_two_level_zscore_statistics(n):
    from scipy import stats
    from hypad_amd.utils import anomaly_detection_utils as adu
    from oracle import scoring as osc
    x = 3.0 + _series(n, n) * 10.0
    got = adu.zscore_clip(x).cpu().numpy()
    if n == 1:
        assert np.isnan(got).all()                                          # 0 / 0, as scipy
        return
    assert np.allclose(got, np.clip(stats.zscore(x), 0, None) + 1, rtol=0, atol=1e-10)
    q = np.quantile(x, [0.25, 0.75])
    c = adu._compute_critic_score(x, 0 if n < 200 else 11).cpu().numpy()
    ref = osc.compute_critic_score(x, 0 if n < 200 else 11)
    assert np.allclose(c, ref, rtol=0, atol=1e-10, equal_nan=True) and q[0] <= q[1]
    y = x.copy(); y[n // 2] = np.nan
    assert np.isnan(adu.zscore_clip(y).cpu().numpy()).all()                 # scipy propagates NaN


def test_score_caches_and_result_table_follow_the_reference(tmp_path, monkeypatch):
    """utils/anomaly_detection_utils.py:470-550 / :225-238 / :112-126 -- with a path the scoring functions leave and re-use the
    reference's artefacts: critic_scores.pickle, point/area/dtw.pickle (z-scored reconstruction scores of ALL three error types),
    anomalies.csv and the results table; a second call reads them back and returns the same scores with empty predictions."""
    import os
    import pickle
    from types import SimpleNamespace
    import pandas as pd
    from hypad_amd.utils import anomaly_detection_utils as adu
    from helpers import load
    fx = load("score.npz")
    y, y_hat, critic = fx["y"], fx["y_hat"], fx["critic"]
    path = str(tmp_path) + "/"
    plain, _, true0, pred0 = adu.score_anomalies(y, y_hat, critic, None, rec_error_type="dtw", comb="mult")
    first, _, _, pred1 = adu.score_anomalies(y, y_hat, critic, None, rec_error_type="dtw", comb="mult", path=path)
    assert np.array_equal(first, plain, equal_nan=True) and len(pred1) == len(pred0)
    assert sorted(os.listdir(path)) == ["area.pickle", "critic_scores.pickle", "dtw.pickle", "point.pickle"]
    again, _, _, pred2 = adu.score_anomalies(y, y_hat, critic, None, rec_error_type="dtw", comb="mult", path=path)
    assert np.array_equal(again, plain, equal_nan=True) and len(pred2) == 0
    point = pickle.load(open(path + "point.pickle", "rb"))
    rec, _ = adu.reconstruction_errors(y, y_hat, 1, 10, len(y) // 100, True, "point", with_summary=False)
    assert np.array_equal(point, adu.zscore_clip(rec).cpu().numpy(), equal_nan=True)
    # the cache is trusted, as in the reference: a doctored critic_scores.pickle shows up in the scores
    cs = pickle.load(open(path + "critic_scores.pickle", "rb"))
    pickle.dump(np.asarray(cs) * 2.0, open(path + "critic_scores.pickle", "wb"))
    doubled, _, _, _ = adu.score_anomalies(y, y_hat, critic, None, rec_error_type="dtw", comb="mult", path=path)
    assert np.allclose(doubled, 2.0 * plain, rtol=1e-12, equal_nan=True)
    # hyperbolic branch: compute_critic_scores re-reads only with params.load; univariate_anomaly_detection writes anomalies.csv + results
    hp = str(tmp_path / "hyper") + "/"
    os.makedirs(hp)
    monkeypatch.chdir(tmp_path)
    P = SimpleNamespace(hyperbolic=True, signal_shape=100, load=False, save_result=True, filename="res.csv", signal="sigA", dataset="synthetic")
    out = adu.univariate_anomaly_detection(fx["ball_recons"], fx["ball_real"], P, "mult", critic, hp, None, signal="sigA", signal_shape=100)
    assert os.path.exists(hp + "critic_scores.pickle") and os.path.exists(hp + "anomalies.csv")
    table = pd.read_csv(tmp_path / "results" / "res.csv")
    assert list(table.columns) == ["signal", "tn", "fp", "fn", "tp"] and list(table["signal"]) == ["sigA"]
    adu.univariate_anomaly_detection(fx["ball_recons"], fx["ball_real"], P, "mult", critic, hp, None, signal="sigA", signal_shape=100)
    assert len(pd.read_csv(tmp_path / "results" / "res.csv")) == 1                      # one row per signal
    pickle.dump(np.full(len(critic) + 99, 3.0), open(hp + "critic_scores.pickle", "wb"))
    P.load = True
    out2 = adu.univariate_anomaly_detection(fx["ball_recons"], fx["ball_real"], P, "mult", critic, hp, None, signal="sigA", signal_shape=100)
    want = 3.0 * adu.hyperbolic_rec_scores(fx["ball_recons"], fx["ball_real"], 100).cpu().numpy().astype(np.float64)
    assert np.allclose(out2["final_scores"], want, rtol=1e-12) and not np.allclose(out["final_scores"], want, rtol=1e-3)


def _quantile_cases():
    rng = np.random.default_rng(11)
    yield "normal", rng.standard_normal(125_099)
    yield "fp32 origin", rng.standard_normal(50_001).astype(np.float32).astype(np.float64)
    yield "narrow band", 0.3 + 1e-9 * rng.standard_normal(70_000)
    yield "heavy ties", rng.integers(0, 4, 40_000).astype(np.float64)
    yield "all equal", np.full(9_999, -2.5)
    yield "signed zeros", np.concatenate([np.zeros(500), -np.zeros(500), rng.standard_normal(31) * 1e-300])
    yield "infinities", np.concatenate([rng.standard_normal(1000), [np.inf] * 400, [-np.inf] * 700])
    yield "wide exponents", rng.standard_normal(30_000) * 10.0 ** rng.integers(-200, 200, 30_000)
    yield "subnormals", rng.standard_normal(3_000) * 5e-324 * 1000
    yield "million", rng.standard_normal(1_000_003)
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 2047, 2049):
        yield "n=%d" % n, rng.standard_normal(n)


def test_device_quantiles_equal_numpy():
    """hypad_quantiles (radix selection on the fp64 keys + numpy's interpolation) against np.quantile: equal bit for bit --
    the quantiles of _compute_critic_score (utils/anomaly_detection_utils.py:319-320) and arbitrary ones."""
    from hypad_amd.utils import anomaly_detection_utils as adu
    for name, x in _quantile_cases():
        for q in ((0.25, 0.75), (0.0, 1.0), (0.5,), (0.1, 0.9), (1.0 / 3.0, 0.999), (0.75, 0.25)):
            got = adu.quantiles(torch.from_numpy(x).cuda(), q).cpu().numpy()
            ref = np.quantile(x, q)
            assert np.array_equal(got, ref, equal_nan=True), (name, q, got, ref)
    x = np.random.default_rng(5).standard_normal(10_000)
    x[1234] = np.nan
    assert np.all(np.isnan(adu.quantiles(x, (0.25, 0.75)).cpu().numpy()))            # numpy: any NaN -> NaN


def test_critic_score_with_device_quantiles_has_the_bits_of_the_host_quantile_form():
    """hypad_critic_score (quantiles taken on the device, read from device memory by the statistics launch) against
    hypad_critic_zscore fed with np.quantile's values: the same bits; and against the NumPy restatement."""
    from hypad_amd import _C
    from hypad_amd.utils import anomaly_detection_utils as adu
    from oracle import scoring as osc
    rng = np.random.default_rng(8)
    for n in (5, 1_000, 125_099):
        modes = rng.standard_normal(n).astype(np.float32).astype(np.float64)
        c = torch.from_numpy(modes).cuda()
        w = max(2, n // 100)
        got = adu._compute_critic_score(c, w).cpu().numpy()
        lo, hi = np.quantile(modes, 0.25), np.quantile(modes, 0.75)
        out = torch.empty_like(c)
        ws = torch.empty(_C.STATS_WORKSPACE_BYTES, dtype=torch.uint8, device="cuda")
        _C.check(_C.lib.hypad_critic_zscore(_C.ptr(c), float(lo), float(hi), _C.ptr(out), n, ws.data_ptr(), _C.STATS_WORKSPACE_BYTES, _C.stream()), "critic_zscore")
        want = adu.rolling_mean(out, w).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True), n
        ref = osc.compute_critic_score(modes, w)
        assert np.allclose(got, ref, rtol=0, atol=1e-9, equal_nan=True), n


def test_unroll_true_from_the_fp32_matrix():
    """hypad_unroll_true_f32 (first column + last row straight from the fp32 window matrix) == the fp64 form."""
    from hypad_amd import _C
    from hypad_amd.utils import anomaly_detection_utils as adu
    rng = np.random.default_rng(2)
    for n, w in ((1, 100), (7, 3), (5_000, 100), (130, 256)):
        y = rng.standard_normal((n, w)).astype(np.float32)
        got = adu.unroll_true(torch.from_numpy(y).cuda()).cpu().numpy()            # fp32 device tensor: the new entry point
        want = adu.unroll_true(y.astype(np.float64)).cpu().numpy()                  # fp64 path
        assert np.array_equal(got, want) and np.array_equal(got, np.concatenate([y[:, 0], y[-1, 1:]]).astype(np.float64)), (n, w)
    series = torch.from_numpy(rng.standard_normal(1_099).astype(np.float32)).cuda()  # row_stride 1: windows of a series
    out = torch.empty(1_099, dtype=torch.float64, device="cuda")
    _C.check(_C.lib.hypad_unroll_true_f32(_C.ptr(series), 1, _C.ptr(out), 1_000, 100, _C.stream()), "unroll_true_f32")
    assert torch.equal(out, series.double())


def test_branches_beside_each_other_equal_one_after_the_other():
    """anomaly_detection_utils.concurrently: the critic smoothing (KDE modes, trimmed z-score, rolling mean) on a side stream beside the
    reconstruction numerics (un-roll median, errors, rolling mean, z-score) == the two run one after the other, bit for bit -- eagerly,
    repeated (the per-stream scratch buffers), and as one replayed graph with the fork and the join as edges; score_anomalies, which
    queues its two halves that way, == its result with the branches forced onto one stream."""
    from hypad_amd import parallel as par
    from hypad_amd.utils import anomaly_detection_utils as adu
    n, S = 6_000, 100
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(n, S, device="cuda", generator=g) * 2 - 1
    y_hat = (x + 0.05 * torch.randn(n, S, device="cuda", generator=g)).contiguous()
    critic = torch.randn(n, device="cuda", generator=g)

    def numerics():
        true = adu.unroll_true(x)
        pred, _ = adu.unroll_predictions(y_hat, False)
        return adu.zscore_clip(adu.rolling_mean(true, 60, minus=pred)), adu.zscore_clip(adu.rolling_mean(adu._dtw_error(true, pred, 10), 60))

    def smoothing():
        return adu._compute_critic_score(adu.kde_modes(critic, S), n // 100)

    a1, a2 = numerics()
    b = smoothing()
    for _ in range(3):
        (c1, c2), d = adu.concurrently(numerics, smoothing)
        torch.cuda.synchronize()
        assert torch.equal(a1, c1) and torch.equal(a2, c2) and torch.equal(b, d)

    def whole():
        (p, q), r = adu.concurrently(numerics, smoothing)
        return p, q, r
    for _ in range(2):
        p, q, r = par.replay_scorer(whole, x, y_hat, critic, key=("beside",))
        torch.cuda.synchronize()
        assert torch.equal(a1, p) and torch.equal(a2, q) and torch.equal(b, r)
    y = x.cpu().numpy()[:, :, None].astype(np.float64)
    got, _, _, _ = adu.score_anomalies(y, y_hat.cpu().numpy(), critic.cpu().numpy(), None, rec_error_type="dtw", comb="mult")
    keep = adu.concurrently
    adu.concurrently = lambda fa, fb: (lambda rb: (fa(), rb))(fb())          # one stream: the side branch first, then the main one
    try:
        want, _, _, _ = adu.score_anomalies(y, y_hat.cpu().numpy(), critic.cpu().numpy(), None, rec_error_type="dtw", comb="mult")
    finally:
        adu.concurrently = keep
    assert np.array_equal(np.asarray(got), np.asarray(want))
