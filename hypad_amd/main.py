"""Command-line driver (reference: main.py:15-70): YAML config -> datasets -> training -> anomaly detection.

    python -m hypad_amd.main --config configs/univariate.yaml [--data-dir ./data] [--resident | --per-iteration]

Default (and ``--drop-in``, kept as an alias): the reference's own call chain -- ``train.train(train_loader, params, config_path)``
over a shuffling ``DataLoader`` (main.py:33-55) with the reference's host random numbers -- where every epoch runs as one
captured ``hypad_train_epoch`` (``train.train_tadgan``, hypad_amd/epoch_feed.py).  ``--per-iteration`` runs that loop call by
call instead (three iteration functions per minibatch); ``--resident`` draws all randomness and the shuffles on the device
(``train.train_resident``: no host work per epoch at all, a different random stream).  Scoring is the same either way: fused
test-loop forward, device scoring kernels, host interval extraction and overlap-segment metrics.  Prints the metrics and
returns them from ``run``."""
import argparse
from types import SimpleNamespace

import numpy as np


def run(params, config_path=None, data_dir="./data", drop_in=True, log=print, resident=False, per_iteration=False):
    import pandas as pd
    from torch.utils.data import DataLoader

    from . import anomaly_detection
    from . import train as ht
    from .utils import anomaly_detection_utils as adu
    from .utils import data as od

    log("dataset: {}, signal: {}".format(params.dataset, params.signal))
    train_dataset, test_dataset, read_path = od.dataset_selection(params, data_dir)
    multivariate = hasattr(train_dataset, "device_windows")            # utils/dataloader_multivariate.py datasets
    if not resident:
        if per_iteration:
            params.per_iteration = True
        train_loader = DataLoader(train_dataset, batch_size=params.batch_size, drop_last=True, shuffle=True, num_workers=0)
        encoder, decoder, critic_x, _, path = ht.train(train_loader, params, config_path)
    else:
        resident = train_dataset.device_windows("cpu") if multivariate else train_dataset
        encoder, decoder, critic_x, _, path, _ = ht.train_resident(resident, params, config_path, log=log)
    test_loader = DataLoader(test_dataset, batch_size=params.batch_size, drop_last=False, shuffle=False, num_workers=0)
    recons_signal, true_signal, critic_score = anomaly_detection.test_tadgan(
        test_loader, encoder, decoder, critic_x, read_path=read_path, signal=params.signal, path=path, signal_shape=params.signal_shape,
        params=params)
    if params.signal == "multivariate" or multivariate:                 # anomaly_detection.py:137-140
        # the reference torch.load()s the labels from its data tree (utils/anomaly_detection_utils.py:143-151); the test
        # dataset already holds that tensor
        y = test_dataset.y if len(getattr(test_dataset, "y", [])) else None
        out = adu.multivariate_anomaly_detection(recons_signal, true_signal, params, params.combination, critic_score, path, y=y)
        log("predicted intervals:\n{}".format(out["intervals"]))
        if out.get("metrics"):
            log("precision: {precision}, recall: {recall}\nf1_score: {f1}, gmean: {gmean}".format(**out["metrics"]))
        return out
    if params.dataset in ("A1", "A2", "A3", "A4"):                       # anomaly_detection.py:32-37
        known = pd.read_csv(read_path[:-4] + "_known_anomalies.csv")
    else:
        known = od.load_anomalies(params.signal, data_dir=data_dir)
    out = adu.univariate_anomaly_detection(recons_signal, true_signal, params, params.combination, critic_score, path, read_path,
                                           params.rec_error, _true_index(test_dataset, params), known, params.signal, params.signal_shape)
    log("predicted intervals:\n{}".format(out["intervals"]))
    log("tn, fp, fn, tp: {}".format(out["confusion"]))
    if out["metrics"]:
        log("precision: {precision}, recall: {recall}\nf1_score: {f1}, gmean: {gmean}".format(**out["metrics"]))
    return out


def _true_index(test_dataset, params):
    """The reference hands the dataset's FULL index to the detector (anomaly_detection.py:127-129: `index[0]`, length N + S): the
    Euclidean branch scores N + S - 1 un-rolled timesteps, so find_anomalies indexes beyond the N window starts.  (The first N
    entries equal X_index, which is all the hyperbolic branch's N window scores need.)"""
    return np.asarray(test_dataset.index)


def main(argv=None):
    import yaml
    ap = argparse.ArgumentParser(description="HypAD on MI355X")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--data-dir", type=str, default="./data")
    ap.add_argument("--drop-in", action="store_true", help="(the default) train.train over a DataLoader with the reference's host-side randomness")
    ap.add_argument("--per-iteration", action="store_true", help="that loop call by call: one critic_x / critic_z / decoder_iteration per minibatch")
    ap.add_argument("--resident", action="store_true", help="device-side randomness and shuffles (train.train_resident)")
    args = ap.parse_args(argv)
    params = SimpleNamespace(**yaml.load(open(args.config), Loader=yaml.FullLoader))
    return run(params, args.config, args.data_dir, resident=args.resident, per_iteration=args.per_iteration)


if __name__ == "__main__":
    main()
