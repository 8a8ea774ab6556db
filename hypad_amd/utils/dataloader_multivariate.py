"""Multivariate window sets (reference: utils/dataloader_multivariate.py:16-121).

``MultivariateDataset`` keeps the reference's constructor, ``X`` / ``y`` attributes and item protocol.  What it does is
load a tensor of rows, view it as ``(-1, 150)`` windows (5 channels x 30 samples, :66), and MinMax-scale every column
to [-1, 1]; the SWaT / WADI branches read a CSV, mean-impute and scale.  Differences, none in the numbers:

* the CSV branches take their files from ``data_dir`` (the reference hard-codes ``./data/SWAT`` and
  ``./data/WADI_downsampled`` relative to the working directory; that is the default);
* scaling and imputation are array expressions in sklearn's order of operations (bit-identical, pinned by
  ``tests/golden/multivariate.npz``) -- no scikit-learn objects are constructed; the ``StratifiedShuffleSplit`` the legacy
  ``CASAS_`` branch draws (:33-34) is never used by the reference and is not reproduced;
* ``device_windows()`` hands the scaled windows to the training / scoring entry points as one resident fp32 matrix.
"""
import os
import sys

import numpy as np

__all__ = ["MultivariateDataset"]


def _minmax_m11(X):
    """sklearn.preprocessing.MinMaxScaler(feature_range=(-1, 1)).fit_transform, column-wise."""
    X = np.asarray(X, dtype=np.float64)
    lo, hi = np.nanmin(X, axis=0), np.nanmax(X, axis=0)
    rng = hi - lo
    rng = np.where(rng == 0.0, 1.0, rng)
    scale = 2.0 / rng
    return X * scale + (-1.0 - lo * scale)


def _impute_mean(X):
    """sklearn.impute.SimpleImputer() (strategy='mean'): NaNs -> column mean; all-NaN columns are dropped."""
    X = np.asarray(X, dtype=np.float64)
    keep = ~np.all(np.isnan(X), axis=0)
    X = X[:, keep]
    mean = np.ma.array(X, mask=np.isnan(X)).mean(axis=0).filled(np.nan)      # sklearn's masked-array mean
    return np.where(np.isnan(X), mean, X)


def _load(path):
    import torch
    t = torch.load(path, weights_only=False)
    return t


# CSV-backed sets: sub-directory, (train file, test file), read_csv keywords, (columns dropped for train, for test)  (:72-108)
_CSV_SETS = {
    "SWAT": ("SWAT", ("SWaT_train_mine.csv", "SWaT_test_mine.csv"), {"index_col": 0},
             (["Timestamp", "Normal/Attack"], ["Timestamp", "Normal/Attack", "label"])),
    "WADI": ("WADI_downsampled", ("WADI_train.csv", "WADI_test_mine.csv"), {}, ([], ["Time", "label"])),
}
_TENSOR_SETS = ("CASAS", "ELINUS", "eHealth")           # one tensor of windows + one of labels, test == train (:65-69)


def _windows(t, width=150):
    """Tensor of samples -> (n, 150) float64 array: 5 channels x 30 samples per window (:66)."""
    return np.asarray(t.reshape(-1, width))


class MultivariateDataset:
    """utils/dataloader_multivariate.py:16-121."""

    def __init__(self, seq_path=None, gt_path=None, test=False, split=1, dataset="CASAS", data_dir="./data"):
        self.test = bool(test)
        which = int(self.test)
        if dataset in _TENSOR_SETS:
            self.X, self.y = _minmax_m11(_windows(_load(seq_path))), _load(gt_path)
        elif dataset == "new_CASAS":                                     # a directory holding x_/y_ train and test (:52-63)
            part = ("train", "test")[which]
            self.X = _minmax_m11(_windows(_load(os.path.join(seq_path, "x_" + part))))
            self.y = _load(os.path.join(seq_path, "y_" + part))
        elif dataset in _CSV_SETS:
            import pandas as pd
            sub, files, kw, drops = _CSV_SETS[dataset]
            frame = pd.read_csv(os.path.join(data_dir, sub, files[which]), **kw).drop(drops[which], axis=1)
            self.X = _minmax_m11(_impute_mean(frame.values))
            self.y = []        # the reference never sets y in these branches, so its test items raise AttributeError
        elif dataset == "CASAS_":
            # legacy two-week recordings (:27-50): rows flattened, the first 4 500 dropped; training = everything up to 1 000
            # rows before the first labelled anomaly, test = from there to 1 000 rows past the last one.  Not rescaled.
            rows = lambda t: t.reshape(t.shape[0] * t.shape[1], -1)[4500:]
            X, y = rows(_load(seq_path)), rows(_load(gt_path))
            hits = np.where(np.asarray(y) == 1)[0]
            lo, hi = hits[0] - 1000, hits[-1] + 1000
            if self.test:
                print("total length: {}, test length: {}, train length: {}".format(y.shape[0], hi - lo, y.shape[0] - (hi - lo)))
            span = slice(lo, hi) if self.test else slice(None, lo)
            self.X, self.y = X[span].reshape(-1, 150), y[span]
        else:
            print("Dataset not supported")
            sys.exit(0)

    def __len__(self):
        return len(self.X)

    def __getitem__(self, idx):
        import torch
        row = self.X[idx]
        x = torch.from_numpy(row) if isinstance(row, np.ndarray) else row
        if self.test:
            return x, [], self.y, [], []
        return x

    # ---- MI355X-native access
    def device_windows(self, device="cuda"):
        """(N, width) fp32 tensor of the scaled windows on the device (one resident matrix for train_epoch / scoring)."""
        import torch
        return torch.as_tensor(np.asarray(self.X), dtype=torch.float32).to(device).contiguous()
