"""Signal / ground-truth files and dataset selection for the univariate configurations (reference: utils/data.py).

``load_anomalies`` (:227-249) and ``dataset_selection`` (:252-379) with the reference's file layout under ``data_dir``
(default ``./data``: ``<signal>.csv``, ``<signal>-train.csv`` / ``<signal>-test.csv``, ``YAHOO/<A?>Benchmark/<signal>.csv``,
``anomalies.csv``).  The S3 download fallback of the reference (:200-224) is not reproduced -- files must be local -- and
the multivariate branches (CASAS / SWAT / WADI tensors that are not part of the reference tree) raise."""
import json
import os

import numpy as np

from .dataloader import SignalDataset

__all__ = ["load_csv", "load_anomalies", "dataset_selection"]


def load_csv(name, data_dir="./data"):
    import pandas as pd
    path = os.path.join(data_dir, name + ".csv")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: the reference downloads missing files from S3 (utils/data.py:200-224); provide it locally")
    return pd.read_csv(path)


def load_anomalies(signal, edges=False, data_dir="./data"):
    """Known anomalous intervals of ``signal`` from ``anomalies.csv`` (utils/data.py:227-249)."""
    import pandas as pd
    table = load_csv("anomalies", data_dir)
    events = table.set_index("signal").loc[signal].values[0]
    anomalies = pd.DataFrame(json.loads(events), columns=["start", "end"])
    if edges:
        data = load_csv(signal, data_dir)
        start, end = data.timestamp.min(), data.timestamp.max()
        anomalies["score"] = 1
        parts = np.concatenate([[[start, anomalies.start.min(), 0]], anomalies.values, [[anomalies.end.max(), end, 0]]], axis=0)
        anomalies = pd.DataFrame(parts, columns=["start", "end", "score"])
    return anomalies


def dataset_selection(params, data_dir="./data"):
    """(train_dataset, test_dataset, read_path) for the univariate branches of utils/data.py:252-379."""
    if params.dataset in ("CASAS_", "new_CASAS", "SWAT", "WADI", "CASAS", "ELINUS", "eHealth"):
        raise NotImplementedError(f"dataset {params.dataset!r}: multivariate tensors are not part of the reference tree")
    if getattr(params, "unique_dataset", False):                       # train == test
        read_path = os.path.join(data_dir, "{}.csv".format(params.signal))
        return (SignalDataset(path=read_path, interval=params.interval),
                SignalDataset(path=read_path, test=True, interval=params.interval), read_path)
    if params.dataset in ("A1", "A2", "A3", "A4"):                     # Yahoo S5
        read_path = os.path.join(data_dir, "YAHOO", "{}Benchmark".format(params.dataset), "{}.csv".format(params.signal))
        return (SignalDataset(path=read_path, interval=1, yahoo=True),
                SignalDataset(path=read_path, test=True, interval=1, yahoo=True), read_path)
    read_path = os.path.join(data_dir, "{}-test.csv".format(params.signal))
    return (SignalDataset(path=os.path.join(data_dir, "{}-train.csv".format(params.signal)), interval=params.interval),
            SignalDataset(path=read_path, interval=params.interval, test=True), read_path)
