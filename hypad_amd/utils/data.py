"""Signal / ground-truth files and dataset selection for the univariate configurations (reference: utils/data.py).

``load_anomalies`` (:227-249) and ``dataset_selection`` (:252-379) with the reference's file layout under ``data_dir``
(default ``./data``: ``<signal>.csv``, ``<signal>-train.csv`` / ``<signal>-test.csv``, ``YAHOO/<A?>Benchmark/<signal>.csv``,
``anomalies.csv``).  The S3 download fallback of the reference (:200-224) is not reproduced -- files must be local -- and
the multivariate branches take the reference's relative file layout from ``data_dir`` too (``DATASETS/<name>/...``, ``SWAT/``,
``WADI_downsampled/``; the tensors themselves are not part of the reference tree).  ``CASAS_`` / ``new_CASAS`` carry the
placeholder roots ``path_to_CASAS`` / ``path_to_new_CASAS`` in the reference (:260,276): here the root is
``params.casas_root`` when given, else that same placeholder."""
import json
import os

import numpy as np

from .dataloader import SignalDataset
from .dataloader_multivariate import MultivariateDataset

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
    """(train_dataset, test_dataset, read_path), utils/data.py:252-379."""
    ds = params.dataset
    if ds == "CASAS_":                                                  # original CASAS, test == train (:259-271)
        root = getattr(params, "casas_root", "path_to_CASAS")
        seq, gt = root + "sequences_2week_{}.pt".format(params.signal), root + "ground_truth_2week_{}.pt".format(params.signal)
        return (MultivariateDataset(seq_path=seq, gt_path=gt, split=params.split, dataset="CASAS_"),
                MultivariateDataset(seq_path=seq, gt_path=gt, test=True, dataset="CASAS_"), "")
    if ds == "new_CASAS":                                               # :274-287
        root = getattr(params, "casas_root", "path_to_new_CASAS") + params.signal
        return (MultivariateDataset(seq_path=root, gt_path=root, split=params.split, dataset=ds),
                MultivariateDataset(seq_path=root, gt_path=root, test=True, dataset=ds), "")
    if ds in ("SWAT", "WADI"):                                          # :289-297
        return MultivariateDataset(dataset=ds, data_dir=data_dir), MultivariateDataset(test=True, dataset=ds, data_dir=data_dir), ""
    if ds in ("CASAS", "ELINUS", "eHealth"):                            # :299-327
        base = os.path.join(data_dir, "DATASETS", ds)
        if not params.new_features:
            seq = os.path.join(base, "normal_sequences.pt")
            seq_test = os.path.join(base, "POINTS", params.signal, "{}_sequences_id{}.pt".format(params.signal, params.id))
            gt = os.path.join(base, "POINTS", params.signal, "{}_groundtruth_id{}.pt".format(params.signal, params.id))
        else:
            seq = os.path.join(base, "normal_sequences_newfeatures.pt")
            seq_test = os.path.join(base, "POINTS_NEWFEATURES", "{}_sequences_newfeatures.pt".format(params.signal))
            gt = os.path.join(base, "POINTS_NEWFEATURES", "{}_groundtruth_newfeatures.pt".format(params.signal))
        return (MultivariateDataset(seq_path=seq, gt_path=gt, split=params.split, dataset=ds),
                MultivariateDataset(seq_path=seq_test, gt_path=gt, test=True, dataset=ds), "")
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
