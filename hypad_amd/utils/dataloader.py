"""Univariate signal -> training windows (reference: utils/dataloader.py).

``SignalDataset`` keeps the reference's constructor, attributes (``X, y, X_index, y_index, index``) and item protocol
(utils/dataloader.py:61-232), so a ``torch.utils.data.DataLoader`` over it behaves as in ``main.py:35-47``.  The steps are
the reference's -- time-bucket mean (:97-137), mean imputation and MinMax scaling to [-1, 1] (:84-87), rolling windows
(:139-222) -- computed with array operations instead of per-bucket / per-window Python loops.

MI355X-native addition: the ``(N, window, 1)`` matrix is a sliding view of the scaled series (window n = series[n : n +
window]), 100x the bytes of what it is made of.  ``device_series()`` puts the series in HBM once and ``window_view()``
describes the windows as rows of that buffer with a row stride of one element: the training and scoring entry points
take that stride (``x_row_stride``, include/hypad.h), so no window matrix is ever materialised on the device.
"""
from datetime import datetime

import numpy as np

__all__ = ["SignalDataset", "time_segments_aggregate", "rolling_window_sequences", "save_known_anomalies", "yahoo_preprocess"]


def time_segments_aggregate(X, interval, time_column, method=("mean",)):
    """Aggregate the value columns of ``X`` (DataFrame or ndarray) over consecutive spans of ``interval`` time units
    starting at the first time stamp (utils/dataloader.py:97-137).  Returns (values (n_segments, n_methods * n_columns),
    first time stamp of each segment); an empty segment aggregates to NaN."""
    import pandas as pd
    if isinstance(X, np.ndarray):
        X = pd.DataFrame(X)
    if isinstance(method, str):
        method = [method]
    X = X.sort_values(time_column).set_index(time_column)
    ts = X.index.values
    vals = X.values.astype(np.float64)
    start, last = ts[0], ts[-1]
    n_seg = int((last - start) // interval) + 1
    seg = ((ts - start) // interval).astype(np.int64)
    index = start + interval * np.arange(n_seg, dtype=ts.dtype if np.issubdtype(ts.dtype, np.integer) else np.float64)
    if list(method) == ["mean"]:                       # the only method the reference's callers use
        # pandas reduces each column as one contiguous vector (NaN -> 0, numpy's pairwise sum, / count of non-NaN):
        # summing the same contiguous slices reproduces its rounding exactly
        bounds = np.searchsorted(seg, np.arange(n_seg + 1), side="left")
        cols_t = np.ascontiguousarray(vals.T)
        ok_t = ~np.isnan(cols_t)
        filled = np.where(ok_t, cols_t, 0.0)
        out = np.full((n_seg, vals.shape[1]), np.nan)
        for b in np.flatnonzero(bounds[1:] > bounds[:-1]):
            lo, hi = bounds[b], bounds[b + 1]
            cnt = ok_t[:, lo:hi].sum(axis=1)
            with np.errstate(invalid="ignore", divide="ignore"):
                out[b] = filled[:, lo:hi].sum(axis=1) / cnt
        return out, index
    # other aggregations: the DataFrame reduction itself, per span (rare path, kept for the signature)
    bounds = np.searchsorted(seg, np.arange(n_seg + 1), side="left")
    out = np.full((n_seg, len(method) * vals.shape[1]), np.nan)
    for b in range(n_seg):
        subset = X.iloc[bounds[b]: bounds[b + 1]]
        out[b] = np.concatenate([getattr(subset, agg)(skipna=True).values for agg in method])
    return out, index


def rolling_window_sequences(X, index, window_size, target_size, step_size, target_column, offset=0, drop=None, drop_windows=False):
    """Input windows, their targets and the first index value of both (utils/dataloader.py:139-222)."""
    X = np.asarray(X)
    index = np.asarray(index)
    target = X[:, target_column]
    max_start = len(X) - window_size - target_size - offset + 1
    if not drop_windows:
        starts = np.arange(0, max(max_start, 0), step_size)
        if starts.size == 0:
            return np.asarray([]), np.asarray([]), np.asarray([]), np.asarray([])
        from numpy.lib.stride_tricks import sliding_window_view
        out_X = sliding_window_view(X, window_size, axis=0)[starts].transpose(0, 2, 1).copy()
        tstart = starts + window_size + offset
        out_y = sliding_window_view(target, target_size)[tstart].copy()
        return out_X, out_y, index[starts], index[tstart]
    if hasattr(drop, "__len__") and not isinstance(drop, str):
        if len(drop) != len(X):
            raise Exception("Arrays `drop` and `X` must be of the same length.")
    elif isinstance(drop, float) and np.isnan(drop):
        drop = np.isnan(X)
    else:
        drop = X == drop
    out_X, out_y, X_index, y_index = [], [], [], []
    start = 0
    while start < max_start:
        end = start + window_size
        bad = np.where(drop[start: end + target_size])[0]
        if bad.size:
            start += bad[-1] + 1
            continue
        out_X.append(X[start:end])
        out_y.append(target[end + offset: end + offset + target_size])
        X_index.append(index[start])
        y_index.append(index[end + offset])
        start += step_size
    return np.asarray(out_X), np.asarray(out_y), np.asarray(X_index), np.asarray(y_index)


def _yahoo_timestamps(n):
    """One time stamp per second from 2012-11-24 local time, the reference's stand-in index (:43-47, :66-75)."""
    t0 = datetime.timestamp(datetime(2012, 11, 24))
    limit = int(datetime.timestamp(datetime(2012, 11, 30)) - t0) + 1
    return t0 + np.arange(min(n, limit), dtype=np.float64)


def save_known_anomalies(df, path):
    """Runs of ``is_anomaly == 1`` as (start, end) time stamps, latest first, written next to ``path`` as
    ``*_known_anomalies.csv`` when ``path`` is given (utils/dataloader.py:14-33).  Returns (df, anomalies)."""
    import pandas as pd
    if "is_anomaly" not in df.columns:
        df = df[["timestamp", "value", "anomaly"]].copy().sort_values(by=["timestamp"])
        df.columns = ["timestamp", "value", "is_anomaly"]
    flag = df["is_anomaly"].values
    ts = df["timestamp"].values
    change = np.flatnonzero(np.concatenate(([True], flag[1:] != flag[:-1])))
    ends = np.concatenate((change[1:], [len(flag)])) - 1
    runs = [(ts[s], ts[e]) for s, e in zip(change, ends) if flag[s] == 1]
    anomalies = pd.DataFrame(runs[::-1], columns=["start", "end"])
    df = df.copy()
    df["csum"] = np.cumsum(np.concatenate(([True], flag[1:] != flag[:-1])))
    if path is not None:
        anomalies.to_csv(path[:-4] + "_known_anomalies.csv")
    return df, anomalies


def yahoo_preprocess(df):
    """Detrend ``value`` and replace the index by the synthetic per-second time stamps (utils/dataloader.py:41-58)."""
    from scipy import signal as scipy_signal
    df = df.copy()
    df["value"] = scipy_signal.detrend(df["value"])
    df["timestamp"] = _yahoo_timestamps(len(df))
    return df[["timestamp", "value"]]


class SignalDataset:
    """utils/dataloader.py:61-232.  ``path`` may also be a DataFrame with ``timestamp`` and value columns."""

    def __init__(self, path, interval=21600, windows_size=100, test=False, yahoo=None, write_known_anomalies=True):
        import pandas as pd
        self.signal_df = pd.read_csv(path) if isinstance(path, str) else path.copy()
        if yahoo:
            from scipy import signal as scipy_signal
            self.signal_df["value"] = scipy_signal.detrend(self.signal_df["value"])
            self.signal_df["timestamp"] = _yahoo_timestamps(len(self.signal_df))
            self.signal_df, self.known_anomalies = save_known_anomalies(
                self.signal_df, path if (write_known_anomalies and isinstance(path, str)) else None)
            self.signal_df = self.signal_df[["timestamp", "value"]]
        self.interval = interval
        self.windows_size = windows_size
        self.test = test
        agg, self.index = time_segments_aggregate(self.signal_df, interval=interval, time_column="timestamp")
        # SimpleImputer(strategy="mean") then MinMaxScaler(feature_range=(-1, 1)), in sklearn's order of operations
        col_mean = np.nanmean(agg, axis=0)
        agg = np.where(np.isnan(agg), col_mean, agg)
        lo, hi = agg.min(axis=0), agg.max(axis=0)
        rng = hi - lo
        rng = np.where(rng == 0.0, 1.0, rng)
        scale = 2.0 / rng
        self.series = agg * scale + (-1.0 - lo * scale)                 # (T, columns), float64
        self.X, self.y, self.X_index, self.y_index = rolling_window_sequences(
            self.series, self.index, window_size=windows_size, target_size=1, step_size=1, target_column=0)
        self._X_built = self.X                                          # (series_windows: X is still the matrix built here)

    def __len__(self):
        return len(self.X)

    def __getitem__(self, idx):
        import torch
        x = torch.from_numpy(self.X[idx])
        if self.test:
            return x, self.index, self.y, self.y_index, self.X_index
        return x

    # ---- MI355X-native access: the series in HBM, windows as overlapping rows of it
    def device_series(self, device="cuda"):
        """(T,) fp32 tensor of the scaled target column on the device (univariate signals)."""
        import torch
        return torch.as_tensor(np.ascontiguousarray(self.series[:, 0]), dtype=torch.float32).to(device)

    def series_windows(self, device="cuda"):
        """window_view() when the window matrix is, row for row, the overlapping rows of the scaled series -- a univariate signal
        whose ``X`` is the object the constructor built (not re-assigned, e.g. to a subset); else None."""
        if self.X is not getattr(self, "_X_built", None) or self.series.shape[1] != 1 or self.X.ndim != 3 or self.X.shape[2] != 1:
            return None
        if len(self.X) + self.windows_size > len(self.series):
            return None
        return self.window_view(device)

    def window_view(self, device="cuda"):
        """(series, n_windows, row_stride = 1): window n is series[n : n + windows_size].  Pass to
        ``Engine.train_epoch(..., x_row_stride=1)`` / ``score_batches`` instead of an (N, window) matrix."""
        return self.device_series(device), len(self.X), 1
