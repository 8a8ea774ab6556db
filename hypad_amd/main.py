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
    """``drop_in`` (default True) = the reference's call chain over a DataLoader; ``drop_in=False`` or ``resident=True`` = the resident path."""
    resident = resident or not drop_in
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
    return _detect(params, test_dataset, read_path, encoder, decoder, critic_x, path, data_dir, multivariate, log)


def _detect(params, test_dataset, read_path, encoder, decoder, critic_x, path, data_dir, multivariate, log):
    """main.py:57-70 / anomaly_detection.py:20-155: the test loop and the detector for one trained model."""
    import pandas as pd
    from torch.utils.data import DataLoader

    from . import anomaly_detection
    from .utils import anomaly_detection_utils as adu
    from .utils import data as od
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


def run_signals(params, names, config_path=None, data_dir="./data", log=print):
    """One model per signal for a list of signals (``--signals a,b,c``): datasets -> ``train.train_signals_resident`` (groups of up
    to 32 models per launch sequence; under ``torchrun`` the signals are sharded over the ranks, one process per GPU) -> per
    signal the test loop and the detector on the rank that trained it -> the metrics of all signals gathered on every rank."""
    import copy

    from . import parallel as par
    from . import train as ht
    from .utils import data as od
    sets = []
    for name in names:
        p = copy.copy(params)
        p.signal = name
        sets.append((p,) + tuple(od.dataset_selection(p, data_dir)))
    trained = ht.train_signals_resident([t[1] for t in sets], params, names=names, log=log)
    local = {}
    for (p, train_ds, test_ds, read_path), name in zip(sets, names):
        mods = trained[name].get("modules")
        if mods is None:
            continue                                   # another rank's signal
        p.latent_space_dim = 20
        out = _detect(p, test_ds, read_path, mods[0], mods[1], mods[2], trained[name]["path"], data_dir, hasattr(train_ds, "device_windows"), log)
        # (tn is None in the overlap-segment count)
        local[name] = {"confusion": [None if v is None else int(v) for v in out.get("confusion", [])] or None, "metrics": out.get("metrics"),
                       "n_intervals": int(len(out["intervals"])), "final": trained[name]["final"], "path": trained[name]["path"],
                       "rank": trained[name]["rank"]}
    return par.gather_signal_metrics(local)


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
    ap.add_argument("--signals", type=str, default=None, help="comma-separated signal names of params.dataset: one model per signal, trained side by "
                                                               "side in groups of up to 32 per GPU (train.train_signals_resident); under torchrun the "
                                                               "signals are sharded over the ranks")
    args = ap.parse_args(argv)
    params = SimpleNamespace(**yaml.load(open(args.config), Loader=yaml.FullLoader))
    if args.signals:
        import os
        import torch
        import torch.distributed as dist
        own_group = False
        if "RANK" in os.environ and not dist.is_initialized():           # one process per GPU (torchrun): RCCL for the end-of-run gather
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            own_group = True
        try:
            res = run_signals(params, [n.strip() for n in args.signals.split(",") if n.strip()], args.config, args.data_dir)
        finally:
            if own_group:
                dist.destroy_process_group()
        for name, r in sorted(res.items()):
            print(name, r["confusion"], r["metrics"])
        return res
    return run(params, args.config, args.data_dir, resident=args.resident, per_iteration=args.per_iteration)


if __name__ == "__main__":
    main()
