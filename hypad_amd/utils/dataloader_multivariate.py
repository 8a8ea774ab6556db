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


class MultivariateDataset:
    """utils/dataloader_multivariate.py:16-121."""

    def __init__(self, seq_path=None, gt_path=None, test=False, split=1, dataset="CASAS", data_dir="./data"):
        import pandas as pd
        self.test = test
        if dataset == "CASAS_":
            self.X = _load(seq_path)
            self.y = _load(gt_path)
            self.X = self.X.reshape(self.X.shape[0] * self.X.shape[1], -1)[4500:]
            self.y = self.y.reshape(self.y.shape[0] * self.y.shape[1], -1)[4500:]
            ynp = np.asarray(self.y)
            init = np.where(ynp == 1)[0][0] - 1000
            end = np.where(ynp == 1)[0][-1] + 1000
            if self.test:
                print("total length: {}, test length: {}, train length: {}".format(self.y.shape[0], end - init,
                                                                                   self.y.shape[0] - (end - init)))
                self.y = self.y[init:end]
                self.X = self.X[init:end].reshape(-1, 150)
            else:
                self.y = self.y[:init]
                self.X = self.X[:init].reshape(-1, 150)
        elif dataset == "new_CASAS":
            part = "test" if self.test else "train"
            self.X = _minmax_m11(np.asarray(_load(os.path.join(seq_path, "x_" + part)).reshape(-1, 150)))
            self.y = _load(os.path.join(seq_path, "y_" + part))
        elif dataset in ("CASAS", "ELINUS", "eHealth"):                  # test == train
            self.X = _minmax_m11(np.asarray(_load(seq_path).reshape(-1, 150)))
            self.y = _load(gt_path)
        elif dataset == "SWAT":
            self.y = []        # the reference never sets y in the SWaT / WADI branches, so its test items raise AttributeError
            if not self.test:
                X = pd.read_csv(os.path.join(data_dir, "SWAT", "SWaT_train_mine.csv"), index_col=0).drop(["Timestamp", "Normal/Attack"], axis=1)
            else:
                X = pd.read_csv(os.path.join(data_dir, "SWAT", "SWaT_test_mine.csv"), index_col=0).drop(["Timestamp", "Normal/Attack", "label"], axis=1)
            self.X = _minmax_m11(_impute_mean(X.values))
        elif dataset == "WADI":
            self.y = []
            if not self.test:
                X = pd.read_csv(os.path.join(data_dir, "WADI_downsampled", "WADI_train.csv"))
            else:
                X = pd.read_csv(os.path.join(data_dir, "WADI_downsampled", "WADI_test_mine.csv")).drop(["Time", "label"], axis=1)
            self.X = _minmax_m11(_impute_mean(X.values))
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
