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
