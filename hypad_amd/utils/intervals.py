"""Anomalous-interval extraction and the overlap-segment confusion matrix.

Host side, O(T) NumPy: the step after the device scoring kernels, on the (T,) score vector they return.  Mirrors
``utils/anomaly_detection_utils.py`` of the reference by name and behaviour:

    find_anomalies                      :1363-1472   sliding windows -> threshold -> sequences -> prune -> score -> merge
    _fixed_threshold / _find_threshold  :1098-1114 / :1066-1095 (z_cost :1023-1063, deltas :965-990, count_above :993-1020)
    _find_sequences                     :1117-1166
    _get_max_errors / _prune_anomalies  :1169-1200 / :1203-1237
    _compute_scores / _merge_sequences  :1240-1269 / :1272-1313
    _overlap / _overlap_segment / contextual_confusion_matrix / compute_metrics   :301-304 / :579-599 / :606-655 / :241-254

The arithmetic is the reference's; the data structures are not (boolean run-length logic on arrays instead of pandas
shift / DataFrame sorting).  Behaviours kept on purpose, because results depend on them:
  * the dynamic threshold keeps the *last* start point whose ``fmin`` cost is finite -- the reference never updates
    ``best_cost`` (:1089-1093);
  * merging weights a sequence by ``stop - start``: two zero-length sequences that touch raise ZeroDivisionError from
    ``np.average`` (:1302), as in the reference;
  * ``weighted=True`` of ``contextual_confusion_matrix`` is not available: the reference names an undefined helper there
    (:635) and fails with NameError; here it is a NotImplementedError.
Pinned by tests/golden/intervals.npz (outputs of the reference's functions).
"""
import numpy as np

__all__ = [
    "find_anomalies", "contextual_confusion_matrix", "compute_metrics", "deltas", "count_above", "z_cost", "casas_anomalies",
]


# ------------------------------------------------------------------------------------------------ thresholds
def _fixed_threshold(errors, k=4):
    """mean + k population standard deviations (:1098-1114)."""
    errors = np.asarray(errors, dtype=np.float64)
    return errors.mean() + k * errors.std()


def deltas(errors, epsilon, mean, std):
    """(mean - mean of the errors <= epsilon, std - their std); (0, 0) when none is below (:965-990)."""
    below = errors[errors <= epsilon]
    if below.size == 0:
        return 0, 0
    return mean - below.mean(), std - below.std()


def count_above(errors, epsilon):
    """Number of errors above epsilon and number of runs of them (:993-1020)."""
    above = np.asarray(errors > epsilon)
    starts = above.copy()
    starts[1:] &= ~above[:-1]
    return int(above.sum()), int(starts.sum())


def z_cost(z, errors, mean, std):
    """Negated goodness of the threshold mean + z std (:1023-1063); inf when nothing is above it."""
    epsilon = mean + z * std
    delta_mean, delta_std = deltas(errors, epsilon, mean, std)
    above, consecutive = count_above(errors, epsilon)
    numerator = -(delta_mean / mean + delta_std / std)
    denominator = above + consecutive ** 2
    if denominator == 0:
        return np.inf
    return numerator / denominator


def _find_threshold(errors, z_range):
    """Dynamic threshold: Nelder-Mead from every integer start point of z_range (:1066-1095)."""
    from scipy.optimize import fmin
    errors = np.asarray(errors, dtype=np.float64)
    mean, std = errors.mean(), errors.std()
    min_z, max_z = z_range
    best_z = min_z
    for z0 in range(min_z, max_z):
        z, cost = fmin(z_cost, z0, args=(errors, mean, std), full_output=True, disp=False)[0:2]
        if cost < np.inf:               # the reference compares with a best_cost it never lowers
            best_z = z[0]
    return mean + best_z * std


# ------------------------------------------------------------------------------------------------ sequences
def _dilate(mask, pad):
    """True within `pad` positions of a True of `mask` (the padding loop of :1146-1150, as a prefix-sum window count)."""
    if pad <= 0 or not mask.any():
        return mask.copy()
    n = mask.size
    c = np.concatenate(([0], np.cumsum(mask, dtype=np.int64)))
    lo = np.clip(np.arange(n) - pad, 0, n)
    hi = np.clip(np.arange(n) + pad + 1, 0, n)
    return (c[hi] - c[lo]) > 0


def _find_sequences(errors, epsilon, anomaly_padding):
    """(start, end) of the padded runs above epsilon, and the largest error outside them (:1117-1166)."""
    errors = np.asarray(errors, dtype=np.float64)
    above = _dilate(errors > epsilon, int(anomaly_padding))
    max_below = 0 if above.all() else errors[~above].max()
    prev = np.concatenate(([False], above[:-1]))
    starts = np.flatnonzero(above & ~prev)
    ends = np.flatnonzero(~above & prev) - 1
    if ends.size == starts.size - 1:
        ends = np.append(ends, above.size - 1)
    return np.stack([starts, ends], axis=1) if starts.size else np.zeros((0, 2), dtype=np.int64), max_below


def _get_max_errors(errors, sequences, max_below):
    """Rows (start, stop, max_error) of every sequence plus the (-1, -1, max_below) sentinel, by descending max_error
    (:1169-1200; a DataFrame there, an (n, 3) array here)."""
    errors = np.asarray(errors, dtype=np.float64)
    rows = [(-1.0, -1.0, float(max_below))]
    rows += [(float(s), float(e), float(errors[int(s): int(e) + 1].max())) for s, e in sequences]
    rows = np.asarray(rows, dtype=np.float64)
    order = np.argsort(-rows[:, 2], kind="stable")
    return rows[order]


def _prune_anomalies(max_errors, min_percent):
    """Keep the leading sequences down to the last one that stands out from its successor by >= min_percent (:1203-1237)."""
    me = np.asarray(max_errors, dtype=np.float64)
    cur, nxt = me[:-1, 2], me[1:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        increase = (cur - nxt) / cur
    big = ~(increase < min_percent)
    if not big.any():
        return me[0:0]
    return me[: int(np.flatnonzero(big)[-1]) + 1]


def _compute_scores(pruned_anomalies, errors, threshold, window_start):
    """[start + window_start, stop + window_start, (max_error - threshold) / (mean + std)] (:1240-1269)."""
    errors = np.asarray(errors, dtype=np.float64)
    denominator = errors.mean() + errors.std()
    return [[row[0] + window_start, row[1] + window_start, (row[2] - threshold) / denominator] for row in pruned_anomalies]


def _merge_sequences(sequences):
    """Merge overlapping or adjacent (start, stop, score) triples; merged score = length-weighted mean (:1272-1313)."""
    if len(sequences) == 0:
        return np.array([])
    ordered = sorted(sequences, key=lambda entry: entry[0])
    merged = [list(ordered[0])]
    scores, weights = [ordered[0][2]], [ordered[0][1] - ordered[0][0]]
    for start, stop, score in ordered[1:]:
        last = merged[-1]
        if start <= last[1] + 1:
            scores.append(score)
            weights.append(stop - start)
            merged[-1] = [last[0], max(last[1], stop), np.average(scores, weights=weights)]
        else:
            scores, weights = [score], [stop - start]
            merged.append([start, stop, score])
    return np.array(merged)


def _find_window_sequences(window, z_range, anomaly_padding, min_percent, window_start, fixed_threshold):
    """Scored anomalous sequences of one window of errors (:1316-1360)."""
    threshold = _fixed_threshold(window) if fixed_threshold else _find_threshold(window, z_range)
    sequences, max_below = _find_sequences(window, threshold, anomaly_padding)
    pruned = _prune_anomalies(_get_max_errors(window, sequences, max_below), min_percent)
    return _compute_scores(pruned, window, threshold, window_start)


def find_anomalies(errors, index, z_range=(0, 10), window_size=None, window_size_portion=None, window_step_size=None,
                   window_step_size_portion=None, min_percent=0.1, anomaly_padding=50, lower_threshold=False,
                   fixed_threshold=None):
    """(index[start], index[stop], score) of every anomalous interval of `errors` (:1363-1472).

    ``univariate_anomaly_detection`` calls it with window_size_portion=0.33, window_step_size_portion=0.1,
    fixed_threshold=True (:89-95).
    """
    errors = np.asarray(errors, dtype=np.float64).reshape(-1)
    n = errors.size
    window_size = window_size or n
    if window_size_portion:
        window_size = int(np.ceil(n * window_size_portion))
    window_step_size = window_step_size or window_size
    if window_step_size_portion:
        window_step_size = int(np.ceil(window_size * window_step_size_portion))
    sequences = []
    window_start = window_end = 0
    while window_end < n:
        window_end = window_start + window_size
        window = errors[window_start:window_end]
        sequences.extend(_find_window_sequences(window, z_range, anomaly_padding, min_percent, window_start, fixed_threshold))
        if lower_threshold:                 # unusually low errors: the window mirrored around its mean
            mean = window.mean()
            sequences.extend(_find_window_sequences(mean - (window - mean), z_range, anomaly_padding, min_percent,
                                                    window_start, fixed_threshold))
        window_start += window_step_size
    return np.asarray([[index[int(start)], index[int(stop)], score] for start, stop, score in _merge_sequences(sequences)])


# ------------------------------------------------------------------------------------------------ metrics
def _overlap(expected, observed):
    """Open-interval intersection test (:301-304)."""
    return (expected[0] - observed[1]) * (expected[1] - observed[0]) < 0


def _as_pairs(intervals):
    if isinstance(intervals, list):
        return [(p[0], p[1]) for p in intervals]
    if hasattr(intervals, "columns"):       # DataFrame with start / end columns
        return list(zip(intervals["start"].tolist(), intervals["end"].tolist()))
    a = np.asarray(intervals)
    return [(r[0], r[1]) for r in a.reshape(-1, a.shape[-1] if a.ndim > 1 else 2)] if a.size else []


def _overlap_segment(expected, observed, start=None, end=None):
    """(None, fp, fn, tp): an expected interval hit by any observed one is one tp; observed intervals that hit nothing are
    fp (:579-599).  Matched observed intervals leave the false-positive pool *by value*, one copy per match, as the
    reference's list.remove does."""
    remaining = list(observed)
    tp = fn = 0
    for ex in expected:
        found = False
        for ob in observed:
            if _overlap(ex, ob):
                found = True
                if ob in remaining:
                    remaining.remove(ob)
        tp += found
        fn += not found
    return None, len(remaining), fn, tp


def contextual_confusion_matrix(expected, observed, data=None, start=None, end=None, weighted=True):
    """(tn, fp, fn, tp) between ground-truth and detected intervals, ends inclusive (:606-655).  Only the overlap-segment
    form (weighted=False, the one ``univariate_anomaly_detection`` uses) exists; tn is None there."""
    if weighted:
        raise NotImplementedError("weighted segments: the reference refers to an undefined _weighted_segment (:635)")
    if data is not None:
        start, end = data["timestamp"].min(), data["timestamp"].max()
    expected = [(a, b + 1) for a, b in _as_pairs(expected)]
    observed = [(a, b + 1) for a, b in _as_pairs(observed)]
    return _overlap_segment(expected, observed, start, end)


def compute_metrics(known_anomalies, pred_anomalies, verbose=True):
    """precision, recall, F1 and their geometric mean from the overlap-segment counts (:241-254; printed there,
    returned as well here)."""
    _, fp, fn, tp = contextual_confusion_matrix(known_anomalies, pred_anomalies, weighted=False)
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.float64(tp) / (tp + fp)
        recall = np.float64(tp) / (tp + fn)
        f1 = 2 * (precision * recall) / (precision + recall)
        gmean = np.sqrt(precision * recall)
    if verbose:
        print("precision: {}, recall: {}".format(precision, recall))
        print("f1_score: {}, gmean: {}".format(f1, gmean))
    return dict(precision=precision, recall=recall, f1=f1, gmean=gmean, tp=tp, fp=fp, fn=fn)


def casas_anomalies(y, x_index):
    """Ground-truth intervals of a 0/1 label tensor y (batches, batch, ...) on the time stamps x_index
    (utils/anomaly_detection_utils.py:279-298).  As in the reference: a run [a, b] of ones is reported as
    (x_index[a], x_index[b - 1]) -- its last point is cut, a one-point run ends before it starts -- and a run that is
    still open at the end of y is dropped.  Returns a DataFrame with columns start, end."""
    import pandas as pd
    x_index = np.asarray(x_index)
    y = np.asarray(y)
    y = y.reshape(y.shape[0] * y.shape[1], -1)[: x_index.shape[0]]
    flag = (y[:, 0] == 1) if y.ndim > 1 else (y == 1)
    prev = np.concatenate(([False], flag[:-1]))
    starts = np.flatnonzero(flag & ~prev)
    ends = np.flatnonzero(~flag & prev)              # first zero after a run: the run is [start, end - 1]
    runs = [(x_index[s], x_index[e - 2]) for s, e in zip(starts, ends)]      # x_index[actual - 1] with actual = e - 1
    return pd.DataFrame.from_records(runs, columns=["start", "end"])
