"""Window-scoring numerics on CPU (oracle; test infrastructure).

NumPy / pandas / SciPy restatement of the scoring half of the hot path,
``utils/anomaly_detection_utils.py`` (SURVEY.md §8a rows S1-S6 and the KDE critic
smoothing of §8f-2).  File I/O, pickling, plotting and interval extraction of the
reference are not part of the path and are not restated.
"""
import math

import numpy as np
import pandas as pd
from scipy import stats


# ----------------------------------------------------------------------------- S1
def unroll_true(y):
    """utils/anomaly_detection_utils.py:908-910 -- first sample of every window plus the
    tail of the last one.  ``y`` is (N, S) or (N, S, 1)."""
    y2 = np.asarray(y).reshape(len(y), -1)
    return np.concatenate([y2[:, 0], y2[-1, 1:]])


def antidiagonal(y_hat, i):
    """Values y_hat[i-j, j] for the valid j of timestep i (:918-921)."""
    n, s = y_hat.shape
    j0, j1 = max(0, i - (n + s - 1) + s), min(i + 1, s)
    j = np.arange(j0, j1)
    return y_hat[i - j, j]


def unroll_predictions(y_hat, with_summary=True):
    """:912-939 -- per-timestep median and [min, p25, p50, p75, max]."""
    y_hat = np.asarray(y_hat)
    n, s = y_hat.shape
    t = n + s - 1
    med = np.empty(t, dtype=y_hat.dtype)
    summ = np.empty((t, 1, 5), dtype=np.float64) if with_summary else None
    for i in range(t):
        v = antidiagonal(y_hat, i)
        med[i] = np.median(v)
        if with_summary:
            summ[i, 0] = [np.min(v), np.percentile(v, 25), np.percentile(v, 50), np.percentile(v, 75), np.max(v)]
    return med, summ


# ------------------------------------------------------------------------- S2 - S4
def point_error(y, y_hat):
    """:761-777"""
    return np.abs(y - y_hat)


def _rolling_apply_centered(x, window, min_periods, fn):
    return pd.Series(x).rolling(window, center=True, min_periods=min_periods).apply(fn, raw=True).values


def area_error(y, y_hat, score_window=10):
    """:780-812.  ``integrate.trapz`` of the reference == ``np.trapezoid`` (unit spacing)."""
    a = _rolling_apply_centered(y, score_window, score_window // 2, np.trapezoid)
    b = _rolling_apply_centered(y_hat, score_window, score_window // 2, np.trapezoid)
    return np.abs(a - b)


def dtw_classic(x, y):
    """pyts.metrics.dtw(x, y) defaults (pyts==0.12.0; PARITY UNPINNED, see package header):
    squared point cost, classic step pattern, square root of the accumulated cost."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    c = (x[:, None] - y[None, :]) ** 2
    d = np.empty_like(c)
    d[0, :] = np.cumsum(c[0, :])
    d[:, 0] = np.cumsum(c[:, 0])
    for i in range(1, len(x)):
        for j in range(1, len(y)):
            d[i, j] = c[i, j] + min(d[i - 1, j], d[i, j - 1], d[i - 1, j - 1])
    return math.sqrt(d[-1, -1])


def dtw_error(y, y_hat, score_window=10):
    """:815-863"""
    length = (score_window // 2) * 2 + 1
    half = length // 2
    yp = np.pad(np.asarray(y, dtype=np.float64), (half, half))
    hp = np.pad(np.asarray(y_hat, dtype=np.float64), (half, half))
    sims = [dtw_classic(yp[i:i + length], hp[i:i + length]) for i in range(max(len(y) - length, 0))]
    return np.asarray([0.0] * half + sims + [0.0] * (len(y) - len(sims) - half))


def rolling_mean_centered(x, window):
    """:953-961 (also :325-330)"""
    return pd.Series(x).rolling(window, center=True, min_periods=window // 2).mean().values


def reconstruction_errors(y, y_hat, score_window=10, smoothing_window=0.01, smooth=True,
                          rec_error_type="point", with_summary=True):
    """:866-962 (step_size fixed at 1 as in the reference's callers)."""
    if isinstance(smoothing_window, float):
        smoothing_window = min(math.trunc(len(y) * smoothing_window), 200)
    true = unroll_true(y)
    pred, summ = unroll_predictions(np.asarray(y_hat), with_summary)
    kind = rec_error_type.lower()
    if kind == "point":
        err = point_error(true, pred)
    elif kind == "area":
        err = area_error(true, pred, score_window)
    elif kind == "dtw":
        err = dtw_error(true, pred, score_window)
    else:
        raise ValueError(rec_error_type)
    if smooth:
        err = rolling_mean_centered(err, smoothing_window)
    return err, summ


def zscore_clip(x):
    """:523-524,542-543 -- ``stats.zscore`` then ``clip(min=0) + 1``."""
    return np.clip(stats.zscore(x), a_min=0, a_max=None) + 1


# ------------------------------------------------------------------ critic smoothing
def kde_mode(v):
    """:380-397 -- the sample at which a Scott-bandwidth Gaussian KDE of ``v`` is largest,
    median fallback when the KDE cannot be built."""
    v = np.asarray(v)
    if len(v) > 1:
        try:
            return v[np.argmax(stats.gaussian_kde(v)(v))]
        except np.linalg.LinAlgError:
            return np.median(v)
    return np.median(v)


def compute_critic_score(critics, smooth_window):
    """:307-333"""
    c = np.asarray(critics)
    lo, hi = np.quantile(c, 0.25), np.quantile(c, 0.75)
    mean = np.mean(c[np.logical_and(c >= lo, c <= hi)])
    z = np.absolute((c - mean) / np.std(c)) + 1
    return rolling_mean_centered(z, smooth_window)


def final_critic_scores(critic_score, n_windows, window):
    """:365-404 -- every window's critic value is repeated along the window, un-rolled
    along anti-diagonals, reduced by the KDE mode and smoothed."""
    ext = np.repeat(np.asarray(critic_score, dtype=np.float64).reshape(-1, 1), window, axis=1)
    t = window + n_windows - 1
    modes = [kde_mode(antidiagonal(ext, i)) for i in range(t)]
    return compute_critic_score(modes, math.trunc(n_windows * 0.01))


# ------------------------------------------------------------------------- S5 / S6
def combine_scores(combination, critic_scores=(), rec_scores=(), recons_signal=()):
    """:336-362 (hyperbolic and multivariate branches)."""
    c, r = np.asarray(critic_scores), np.asarray(rec_scores)
    if combination in ("uncertainty", "critic_uncertainty", "sum_uncertainty", "rec_uncertainty"):
        unc = np.linalg.norm(np.asarray(recons_signal), axis=1)
    if combination == "sum":
        return 0.2 * c + 0.8 * r
    if combination == "mult":
        return c * r
    if combination == "uncertainty":
        return c * r * unc
    if combination == "critic":
        return c
    if combination == "critic_uncertainty":
        return c * unc
    if combination == "sum_uncertainty":
        return 0.5 * c * unc[: r.shape[0]] + 0.5 * r * unc[: r.shape[0]]
    if combination == "rec":
        return r
    if combination == "rec_uncertainty":
        return r * unc
    raise ValueError(combination)


def combine_euclidean(comb, critic_scores, rec_scores, lambda_rec=0.5):
    """:553-570 (score_anomalies tail)."""
    if comb == "mult":
        return np.multiply(critic_scores, rec_scores)
    if comb == "sum":
        return (1 - lambda_rec) * (critic_scores - 1) + lambda_rec * (rec_scores - 1)
    if comb == "rec":
        return rec_scores
    if comb == "critic":
        return critic_scores
    raise ValueError(comb)


def score_anomalies(y, y_hat, critic, rec_error_type="point", comb="mult", score_window=10):
    """:407-576 without the pickle caches: KDE critic scores, reconstruction scores
    (z-scored, clipped, +1) and their combination."""
    n = y.shape[0]
    w = math.trunc(n * 0.01)
    critic_scores = final_critic_scores(critic, y_hat.shape[0], y_hat.shape[1])
    rec, _ = reconstruction_errors(y, y_hat, score_window, w, True, rec_error_type, with_summary=False)
    rec = zscore_clip(rec)
    return combine_euclidean(comb, critic_scores, rec), critic_scores, rec


def hyperbolic_scores(recons, real_hyper, critic, combination="mult"):
    """:54-86 -- row-wise Poincare distance, KDE critic scores, combination."""
    import torch
    from . import gmath
    a = torch.as_tensor(np.asarray(real_hyper), dtype=torch.float32)
    b = torch.as_tensor(np.asarray(recons), dtype=torch.float32)
    rec = gmath.rowwise_poincare_distance(a, b)
    crit = final_critic_scores(critic, recons.shape[0], recons.shape[1])[: rec.shape[0]]
    return combine_scores(combination, crit, rec.numpy(), recons), crit, rec.numpy()
