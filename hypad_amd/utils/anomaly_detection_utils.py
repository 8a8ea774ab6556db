"""Window scoring on the GPU (reference: utils/anomaly_detection_utils.py).

Same function names and argument meaning as the reference for the numerics on the hot path
(SURVEY.md §8a rows S1-S6); inputs may be NumPy arrays (as the reference passes) or device tensors, results
come back as NumPy arrays like the reference's.  Interval extraction and the overlap-segment metrics
 are host-side NumPy in ``hypad_amd.utils.intervals`` and re-exported here under the reference's
names.  The reference's cached artefacts are kept under their names and formats -- ``critic_scores.pickle``,
``point.pickle`` / ``area.pickle`` / ``dtw.pickle`` (:229-235, :470-550), ``anomalies.csv`` (:94-95) and the results table
``./results/<params.filename>`` (:115-126) -- whenever a ``path`` is given; plotting is not part of this module.
"""
import ctypes
import math
import os
import pickle

import numpy as np
import torch

from .. import _C
from ..hyperspace import gmath
from .intervals import (_find_sequences, _find_threshold, _fixed_threshold, _merge_sequences, _overlap, _prune_anomalies,  # noqa: F401
                        casas_anomalies, compute_metrics, contextual_confusion_matrix, find_anomalies)


def _dev():
    return torch.device("cuda")


def _f32(a):
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    return t.to(_dev(), torch.float32).contiguous()


def _f64(a):
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    return t.to(_dev(), torch.float64).contiguous()


def unroll_true(y):
    """First sample of every window plus the tail of the last (:908-910).  y: (N, S) or (N, S, 1)."""
    if isinstance(y, torch.Tensor) and y.is_cuda and y.dtype == torch.float32 and y.is_contiguous():
        y = y.reshape(y.shape[0], -1)               # the matrix the forward read: take the n + S - 1 values straight from it
        n, w = y.shape
        out = torch.empty(n + w - 1, device=y.device, dtype=torch.float64)
        _C.check(_C.lib.hypad_unroll_true_f32(_C.ptr(y), w, _C.ptr(out), n, w, _C.stream()), "unroll_true_f32")
        return out
    y = _f64(y)
    y = y.reshape(y.shape[0], -1)
    n, w = y.shape
    out = torch.empty(n + w - 1, device=y.device, dtype=torch.float64)
    _C.check(_C.lib.hypad_unroll_true(_C.ptr(y), _C.ptr(out), n, w, _C.stream()), "unroll_true")
    return out


def unroll_predictions(y_hat, with_summary=True):
    """Per-timestep median (float32) and [min, p25, p50, p75, max] over the anti-diagonals (:918-935)."""
    y_hat = _f32(y_hat)
    n, w = y_hat.shape
    t = n + w - 1
    med = torch.empty(t, device=y_hat.device, dtype=torch.float32)
    summ = torch.empty(t, 5, device=y_hat.device, dtype=torch.float64) if with_summary else None
    _C.check(_C.lib.hypad_unroll_median(_C.ptr(y_hat), _C.ptr(med), _C.ptr(summ), n, w, _C.stream()), "unroll_median")
    return med, summ


def _point_wise_error(y, y_hat):
    y, y_hat = _f64(y), _f32(y_hat)
    out = torch.empty_like(y)
    _C.check(_C.lib.hypad_point_error(_C.ptr(y), _C.ptr(y_hat), _C.ptr(out), y.numel(), _C.stream()), "point_error")
    return out


def _area_error(y, y_hat, score_window=10):
    y, y_hat = _f64(y), _f32(y_hat)
    out = torch.empty_like(y)
    _C.check(_C.lib.hypad_area_error(_C.ptr(y), _C.ptr(y_hat), _C.ptr(out), y.numel(), score_window, _C.stream()), "area_error")
    return out


def _dtw_error(y, y_hat, score_window=10):
    y, y_hat = _f64(y), _f32(y_hat)
    out = torch.empty_like(y)
    _C.check(_C.lib.hypad_dtw_error(_C.ptr(y), _C.ptr(y_hat), _C.ptr(out), y.numel(), score_window, _C.stream()), "dtw_error")
    return out


_SCRATCH = {}


def _scratch(device, nbytes, tag):
    """A scratch buffer for the library calls' workspaces, per device AND stream (grown, never shrunk; stream-ordered like everything
    here -- two branches of a pass that run beside each other on two streams, `concurrently`, must not share one)."""
    key = (str(device), tag, _C.stream().value if torch.device(device).type == "cuda" else None)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _SCRATCH[key] = torch.empty(max(int(nbytes), 64), dtype=torch.uint8, device=device)
    return buf


_SIDE_STREAMS = {}


def _side_stream_for(main, dev):
    """A stream whose work really runs beside ``main``'s (hypad_amd/streams.py: streams can share a hardware queue)."""
    from .. import streams
    return streams.beside([main], dev)


def concurrently(fn_main, fn_side):
    """``fn_main()`` on the current stream and ``fn_side()`` on a side stream BESIDE it: both start behind everything queued so far,
    whatever is queued afterwards starts behind both.  Returns (fn_main(), fn_side()).  What it is for: the critic smoothing of a
    scoring pass is one chip-filling kernel (the KDE modes, 0.26 ms per 125 000 windows) plus ~10 launch-sized ones, the
    reconstruction numerics are ~12 launch-sized kernels (0.17 ms): independent of each other, both read only the forward's outputs,
    and beside each other the small launches disappear under the large one.  Capturable (the fork and the join become graph edges)."""
    dev = torch.cuda.current_device()
    main = torch.cuda.current_stream()
    side = _SIDE_STREAMS.get((dev, main.cuda_stream))
    if side is None:
        side = _SIDE_STREAMS[(dev, main.cuda_stream)] = _side_stream_for(main, dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b = fn_side()
    a = fn_main()
    main.wait_stream(side)
    return a, b


def rolling_mean(x, window, origin=0, minus=None):
    """pandas ``rolling(window, center=True, min_periods=window // 2).mean()`` (:953-961, :325-330).  ``window == 0`` -- the
    reference's ``math.trunc(n * 0.01)`` for fewer than 100 windows -- gives all-NaN, as pandas does.  ``minus``: smooth the
    point-wise error ``|x - minus|`` (:761-777) without materialising it.  ``origin``: position of ``x[0]`` in the whole series
    when ``x`` is a slice of it (the sums are taken in an order fixed by absolute positions: a slice gives the whole's bits)."""
    x = _f64(x)
    if int(window) == 0:
        return torch.full_like(x, float("nan"))
    out = torch.empty_like(x)
    sub = None if minus is None else _f32(minus)
    nbytes = _C.lib.hypad_rolling_workspace_bytes(x.numel())
    ws = _scratch(x.device, nbytes, "roll")
    _C.check(_C.lib.hypad_rolling_mean(_C.ptr(x), _C.ptr(sub), _C.ptr(out), x.numel(), int(window), int(origin), ws.data_ptr(), nbytes,
                                       _C.stream()), "rolling_mean")
    return out


def zscore_clip(x):
    """stats.zscore(x) -> clip(min=0) + 1  (:523-524)."""
    x = _f64(x)
    out = torch.empty_like(x)
    ws = _scratch(x.device, _C.STATS_WORKSPACE_BYTES, "stats")
    _C.check(_C.lib.hypad_zscore_clip(_C.ptr(x), _C.ptr(out), x.numel(), ws.data_ptr(), _C.STATS_WORKSPACE_BYTES, _C.stream()), "zscore_clip")
    return out


def reconstruction_errors(y, y_hat, step_size=1, score_window=10, smoothing_window=0.01, smooth=True, rec_error_type="point",
                          with_summary=True):
    """:866-962.  Returns (errors, predictions_vs) as NumPy arrays like the reference."""
    if step_size != 1:
        raise NotImplementedError("step_size != 1 (the reference's callers always use 1)")
    n = len(y)
    if isinstance(smoothing_window, float):
        smoothing_window = min(math.trunc(n * smoothing_window), 200)
    true = unroll_true(y)
    pred, summ = unroll_predictions(y_hat, with_summary)
    kind = rec_error_type.lower()
    if kind == "point" and smooth and smoothing_window:
        err = rolling_mean(true, smoothing_window, minus=pred)         # |true - pred| smoothed in one pass
        smooth = False
    elif kind == "point":
        err = _point_wise_error(true, pred)
    elif kind == "area":
        err = _area_error(true, pred, score_window)
    elif kind == "dtw":
        err = _dtw_error(true, pred, score_window)
    else:
        raise ValueError(rec_error_type)
    if smooth:
        err = rolling_mean(err, smoothing_window)
    pvs = summ.cpu().numpy().reshape(-1, 1, 5) if summ is not None else np.empty((0, 1, 5))
    return err.cpu().numpy(), pvs


def hyperbolic_rec_scores(recons_signal, true_signal, signal_shape):
    """Row-wise Poincare distance between real windows on the ball and reconstructions (:54-66)."""
    true_data = _f32(recons_signal).reshape(-1, signal_shape)
    pred_data = _f32(true_signal).reshape(-1, signal_shape)
    return gmath.poincare_rowdist(pred_data, true_data)


def row_norms(x):
    x = _f32(x)
    out = torch.empty(x.shape[0], device=x.device, dtype=torch.float64)
    _C.check(_C.lib.hypad_row_norms(_C.ptr(x), _C.ptr(out), x.shape[0], x.shape[1], _C.stream()), "row_norms")
    return out


def combine_scores(combination, critic_scores=[], rec_scores=[], recons_signal=[], norms=None, as_tensor=False):
    """:336-362.  ``norms``: the row norms of ``recons_signal`` when the caller already has them (sharded scoring);
    ``as_tensor``: leave the result on the device instead of returning NumPy."""
    if combination not in ("sum", "mult", "uncertainty", "critic", "critic_uncertainty", "sum_uncertainty", "rec", "rec_uncertainty"):
        raise ValueError(combination)
    c = _f64(critic_scores) if len(critic_scores) else None
    r = _f64(rec_scores) if len(rec_scores) else None
    n = (r if r is not None else c).numel()
    u = None
    if "uncertainty" in combination:
        u = (_f64(norms) if norms is not None else row_norms(recons_signal))[:n].contiguous()
    if c is not None:
        c = c[:n].contiguous()
    out = torch.empty(n, device=_dev(), dtype=torch.float64)
    _C.check(_C.lib.hypad_combine_scores(_C.COMB[combination], _C.ptr(c), _C.ptr(r), _C.ptr(u), _C.ptr(out), n, _C.stream()), "combine")
    return out if as_tensor else out.cpu().numpy()


def combine_euclidean(comb, critic_scores, rec_scores, as_tensor=False):
    """Tail of score_anomalies (:553-570), lambda_rec = 0.5."""
    mode = {"mult": "eucl_mult", "sum": "eucl_sum", "rec": "rec", "critic": "critic"}.get(comb)
    if mode is None:
        raise ValueError('Unknown combination specified {}, use "mult", "sum", or "rec" instead.'.format(comb))
    c, r = _f64(critic_scores), _f64(rec_scores)
    out = torch.empty_like(r)
    _C.check(_C.lib.hypad_combine_scores(_C.COMB[mode], _C.ptr(c), _C.ptr(r), None, _C.ptr(out), r.numel(), _C.stream()), "combine")
    return out if as_tensor else out.cpu().numpy()


def quantiles(x, q):
    """``np.quantile(x, q)`` (method "linear") of a device vector for one or two ``q``: exact order statistics by radix
    selection (hypad_quantiles), the result stays on the device -- (len(q),) float64."""
    c = _f64(x).reshape(-1)
    q = [float(v) for v in np.atleast_1d(q)]
    qa = (ctypes.c_double * len(q))(*q)
    out = torch.empty(len(q), dtype=torch.float64, device=c.device)
    nbytes = _C.lib.hypad_quantile_workspace_bytes()
    ws = _scratch(c.device, nbytes, "quantiles")
    _C.check(_C.lib.hypad_quantiles(_C.ptr(c), c.numel(), qa, len(q), _C.ptr(out), ws.data_ptr(), nbytes, _C.stream()), "quantiles")
    return out


def _compute_critic_score(critics, smooth_window):
    """:307-333 -- quantile-trimmed mean, |z| + 1, centred rolling mean.  All on the device: the two quantiles by radix
    selection (no sort), one reduction kernel, one elementwise kernel, the rolling mean; nothing passes through the host."""
    c = _f64(critics)
    out = torch.empty_like(c)
    nbytes = _C.lib.hypad_critic_score_workspace_bytes()
    ws = _scratch(c.device, nbytes, "critic_score")
    _C.check(_C.lib.hypad_critic_score(_C.ptr(c), _C.ptr(out), c.numel(), ws.data_ptr(), nbytes, _C.stream()), "critic_score")
    return rolling_mean(out, smooth_window)


def kde_modes(critic_score, window):
    """Per un-rolled timestep, the KDE mode of the covering windows' critic values (:374-400)."""
    c = _f32(critic_score).reshape(-1)
    n = c.numel()
    modes = torch.empty(n + window - 1, device=c.device, dtype=torch.float64)
    _C.check(_C.lib.hypad_kde_mode(_C.ptr(c), _C.ptr(modes), n, int(window), _C.stream()), "kde_mode")
    return modes


def final_critic_scores(critic_score, true_signal):
    """:365-404."""
    n, w = true_signal.shape[0], true_signal.shape[1]
    return _compute_critic_score(kde_modes(critic_score, w), math.trunc(n * 0.01)).cpu().numpy()


def _load_pickle(file):
    with open(file, "rb") as handle:
        return pickle.load(handle)


def _dump_pickle(obj, file):
    with open(file, "wb") as handle:
        pickle.dump(obj, handle, protocol=pickle.HIGHEST_PROTOCOL)


def compute_critic_scores(rec_scores, critic_score, true_signal, params, path):
    """:225-238 -- the smoothed KDE critic scores of the hyperbolic / multivariate branches, cached as
    ``path + "critic_scores.pickle"`` (read back only when ``params.load`` is set, always re-written otherwise)."""
    file = (path or "") + "critic_scores.pickle"
    if path and getattr(params, "load", False) and os.path.exists(file):
        critic_scores = np.asarray(_load_pickle(file))
    else:
        ts = np.asarray(true_signal)
        critic_scores = final_critic_scores(critic_score, ts.reshape(len(ts), -1))
        if path:
            _dump_pickle(critic_scores, file)
    return critic_scores[: len(rec_scores)]


def score_anomalies(y, y_hat, critic, index=None, score_window=10, critic_smooth_window=None, error_smooth_window=None,
                    smooth=True, rec_error_type="point", comb="mult", lambda_rec=0.5, path=None, samples_num="0", with_true=True):
    """:407-576.  Returns (final_scores, true_index, true, predictions) (``with_true=False``: ``true`` = [], see below).  With a ``path`` the reference's caches are kept:
    ``critic_scores.pickle`` is read if present (else computed and written); the z-scored reconstruction scores of all three
    error types are written as ``point.pickle`` / ``area.pickle`` / ``dtw.pickle`` when missing, and the requested one is read
    back if it was there already (``predictions`` is then empty, as in the reference)."""
    if lambda_rec != 0.5:
        raise NotImplementedError("lambda_rec != 0.5")
    n = y.shape[0]
    critic_smooth_window = critic_smooth_window or math.trunc(n * 0.01)
    error_smooth_window = error_smooth_window or math.trunc(n * 0.01)
    cfile = (path or "") + "critic_scores.pickle"
    cached_critic = bool(path) and os.path.exists(cfile)

    def critic_branch():
        if cached_critic:
            return np.asarray(_load_pickle(cfile))
        return _compute_critic_score(kde_modes(critic, y_hat.shape[1]), critic_smooth_window)

    def rec_scores_of(kind):
        rec, predictions = reconstruction_errors(y, y_hat, 1, score_window, error_smooth_window, smooth, kind)
        return zscore_clip(rec), predictions

    def rec_branch():
        had_requested = bool(path) and os.path.exists(path + rec_error_type + ".pickle")
        if path:
            for kind in ("point", "area", "dtw"):
                if not os.path.exists(path + kind + ".pickle"):
                    _dump_pickle(rec_scores_of(kind)[0].cpu().numpy(), path + kind + ".pickle")
        if had_requested:
            return np.asarray(_load_pickle(path + rec_error_type + ".pickle")), []
        rec_scores, predictions = rec_scores_of(rec_error_type)
        if path:
            _dump_pickle(rec_scores.cpu().numpy(), path + rec_error_type + ".pickle")
        return rec_scores, predictions

    # the critic smoothing (queued first, on a side stream) runs beside the reconstruction errors (this stream, with the host round
    # trips the reference's NumPy return types ask for)
    if torch.cuda.is_available() and not cached_critic:
        (rec_scores, predictions), critic_scores = concurrently(rec_branch, critic_branch)
    else:
        critic_scores = critic_branch()
        rec_scores, predictions = rec_branch()
    if path and not cached_critic:
        _dump_pickle(critic_scores.cpu().numpy(), cfile)
    final = combine_euclidean(comb, critic_scores, rec_scores)
    # [[t0], [t1], ...] as the reference returns it: 250 000 Python objects at 125 000 windows, 50 ms -- two thirds of the whole detector
    # call; ``with_true=False`` (univariate_anomaly_detection, which drops it like the reference's caller does) returns [] instead
    true = unroll_true(y).cpu().numpy().astype(np.float64).reshape(-1, 1).tolist() if with_true else []
    return final, index, true, predictions


def hyperbolic_scores(recons_signal, true_signal, critic_score, signal_shape, combination="mult", params=None, path=None):
    """The hyperbolic branch of univariate_anomaly_detection (:54-86) up to final_scores (``params`` / ``path``: the
    ``critic_scores.pickle`` cache of compute_critic_scores)."""
    rec = hyperbolic_rec_scores(recons_signal, true_signal, signal_shape)
    critic_scores = []
    if combination in ("mult", "uncertainty", "sum", "sum_uncertainty", "critic", "critic_uncertainty"):
        critic_scores = compute_critic_scores(rec, critic_score, true_signal, params, path)
    return combine_scores(combination, critic_scores, rec, recons_signal)


def save_result(params, signal, out):
    """:112-126 -- one row [signal, tn, fp, fn, tp] appended to ``./results/<params.filename>`` unless ``params.signal`` already
    has one (the reference's check), the table created with its header when missing."""
    import pandas as pd
    file_place = "./results/{}".format(params.filename)
    os.makedirs(os.path.dirname(file_place), exist_ok=True)
    res = pd.read_csv(file_place) if os.path.isfile(file_place) else pd.DataFrame(columns=["signal", "tn", "fp", "fn", "tp"])
    if params.signal not in list(res["signal"]):
        res.loc[len(res)] = [signal] + list(out)
        res.to_csv(file_place, index=False)
    return file_place


def univariate_anomaly_detection(recons_signal, true_signal, params, combination, critic_score, path=None, read_path=None,
                                 rec_error_type="euclidean", true_index=None, known_anomalies=None, signal=None,
                                 signal_shape=None):
    """:21-127 end to end: window scores on the device, interval extraction and overlap-segment counts on the host.

    The reference's artefacts are written when ``path`` is given -- the score caches (score_anomalies / compute_critic_scores),
    ``path + "anomalies.csv"`` with the predicted intervals, and, with ``params.save_result``, the results table
    (``save_result``).  ``read_path`` is accepted and ignored (the reference reads that CSV only for its timestamp range,
    which the overlap form of the metrics does not use).  The result is returned instead of printed:
        dict(final_scores, intervals (n, 3) [start, end, score], confusion [tn, fp, fn, tp] or [0, 0, 0, 0] when no
        interval was predicted (the reference's except branch), metrics or None)
    """
    if not params.hyperbolic:
        final_scores, true_index, _, _ = score_anomalies(true_signal, recons_signal, critic_score, true_index,
                                                         rec_error_type=rec_error_type, comb=combination, path=path, with_true=False)
    else:
        final_scores = hyperbolic_scores(recons_signal, true_signal, critic_score, params.signal_shape, combination, params, path)
    final_scores = np.asarray(final_scores, dtype=np.float64).reshape(-1)
    if true_index is None:
        true_index = np.arange(final_scores.size)
    intervals = find_anomalies(final_scores, true_index, window_size_portion=0.33, window_step_size_portion=0.1,
                               fixed_threshold=True)
    out = dict(final_scores=final_scores, intervals=np.asarray(intervals, dtype=np.float64).reshape(-1, 3), confusion=[0, 0, 0, 0],
               metrics=None)
    if path:
        import pandas as pd
        pd.DataFrame(out["intervals"], columns=["start", "end", "score"]).to_csv(path + "anomalies.csv")
    if known_anomalies is not None and out["intervals"].shape[0] > 0:
        pred = [(r[0], r[1]) for r in out["intervals"]]
        out["confusion"] = list(contextual_confusion_matrix(known_anomalies, pred, weighted=False))
        out["metrics"] = compute_metrics(known_anomalies, pred, verbose=False)
    if getattr(params, "save_result", False):
        out["results_file"] = save_result(params, signal, [int(v) for v in out["confusion"]])
    return out


def multivariate_anomaly_detection(recons_signal, true_signal, params, combination, critic_score, path=None, y=None, x_index=None):
    """:129-222 with the ground truth passed in instead of loaded from the reference's data tree (``y``: the 0/1 label
    tensor the reference torch.load()s, or None) and nothing written or plotted.  Reconstruction score: z-scored L2 norm
    (Euclidean) or z-scored row-wise Poincare distance (hyperbolic), on the device; intervals with the multivariate
    settings (window 0.2 T, step 0.1 window, padding 200).  Returns dict(final_scores, intervals, known_anomalies, metrics)."""
    n = len(recons_signal)
    if x_index is None:
        from .dataloader import _yahoo_timestamps       # the reference's stand-in index: one time stamp per second (:133-137)
        x_index = _yahoo_timestamps(n)
    if not params.hyperbolic:
        diff = _f32(np.asarray(true_signal, dtype=np.float32).reshape(n, -1) - np.asarray(recons_signal, dtype=np.float32).reshape(n, -1))
        rec = row_norms(diff)
    else:
        rec = hyperbolic_rec_scores(recons_signal, true_signal, params.signal_shape)
    rec = rec if isinstance(rec, torch.Tensor) else torch.as_tensor(np.asarray(rec, dtype=np.float64))
    rec_scores = zscore_clip(rec.to(torch.float64)).cpu().numpy()
    critic_scores = []
    if combination in ("mult", "uncertainty", "sum", "sum_uncertainty", "critic", "critic_uncertainty"):
        ts = np.asarray(true_signal)
        critic_scores = final_critic_scores(critic_score, ts.reshape(len(ts), -1))[: rec_scores.shape[0]]
    final_scores = np.asarray(combine_scores(combination, critic_scores, rec_scores, recons_signal), dtype=np.float64).reshape(-1)
    intervals = find_anomalies(final_scores, x_index, window_size_portion=0.2, window_step_size_portion=0.1, fixed_threshold=True,
                               anomaly_padding=200)
    out = dict(final_scores=final_scores, intervals=np.asarray(intervals, dtype=np.float64).reshape(-1, 3), known_anomalies=None, metrics=None)
    if y is not None:
        known = casas_anomalies(y, np.asarray(x_index))
        out["known_anomalies"] = known
        if out["intervals"].shape[0] and len(known):
            out["metrics"] = compute_metrics(known, [(r[0], r[1]) for r in out["intervals"]], verbose=False)
    return out
