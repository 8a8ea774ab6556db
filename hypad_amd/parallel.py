"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL.

The reference is single-process; what shards is the *unit of work* it already treats independently:

* training -- one model per signal (train.py:428-437): rank r owns signals {s : s mod world == r}; nothing is
  exchanged while training (no gradient all-reduce exists, each signal has its own weights).  Only end-of-run
  metrics are gathered.
* scoring -- windows are independent through the networks; the un-roll over anti-diagonals needs a halo of
  (window - 1) windows at each boundary, which is re-computed locally instead of exchanged; the global z-score
  needs one all-reduce of (sum, sum of squares, count).
"""
import torch
import torch.distributed as dist


def _group_active(group=None):
    """True when a process group exists: the collectives below then RUN -- also at world size 1 (a one-GPU box still creates
    the RCCL communicator and executes broadcast / all-gather / all-reduce, so the nccl branches are the code that is
    tested, not dead code until an 8-GPU node shows up).  Without a process group they are no-ops."""
    return dist.is_available() and dist.is_initialized()


def signals_of_rank(n_signals, world, rank):
    """Round-robin ownership of signals."""
    return list(range(rank, n_signals, world))


def window_range(n_windows, world, rank):
    """Contiguous, balanced [begin, end) window range of a rank."""
    base, extra = divmod(n_windows, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def window_range_with_halo(n_windows, world, rank, window):
    """Range a rank must *evaluate* so that every un-rolled timestep it owns sees all its contributing windows
    (timestep t gathers y_hat[t - j, j], j < window: utils/anomaly_detection_utils.py:918-921)."""
    b, e = window_range(n_windows, world, rank)
    return max(0, b - (window - 1)), e


def timestep_range(n_windows, world, rank, window):
    """Un-rolled timesteps owned by a rank: those whose newest contributing window is in its window range
    (the last rank also owns the tail of the final window)."""
    b, e = window_range(n_windows, world, rank)
    return b, (e if rank < world - 1 else n_windows + window - 1)


def global_zscore_stats(local_sum, local_sumsq, local_count, group=None):
    """Mean and population std over all ranks from one all-reduce of (sum, sum sq, count) (stats.zscore, ddof=0)."""
    t = torch.tensor([local_sum, local_sumsq, local_count], dtype=torch.float64)
    if _group_active(group):
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t.cpu()
    s, ss, n = t.tolist()
    mean = s / n
    var = max(ss / n - mean * mean, 0.0)
    return mean, var ** 0.5


def broadcast_weights(modules, src=0, group=None):
    """One-time broadcast of the shared generator / critic weights for sharded scoring (~1 MB, RCCL
    broadcast over xGMI): every module's flat parameter arena is overwritten with rank ``src``'s.  A no-op without a
    process group.  Returns the number of bytes broadcast."""
    if not _group_active(group):
        return 0
    nbytes = 0
    for m in modules:
        arena = m.arena()
        if dist.get_backend(group) == "nccl":
            dist.broadcast(arena, src, group=group)
        else:
            host = arena.cpu()
            dist.broadcast(host, src, group=group)
            arena.copy_(host)
        nbytes += arena.numel() * arena.element_size()
    return nbytes


def gather_signal_metrics(local, group=None):
    """{signal_id: metrics} from every rank -> merged dict on every rank (end-of-run only; KB-sized)."""
    if not _group_active(group):
        return dict(local)
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, dict(local), group=group)
    merged = {}
    for d in out:
        merged.update(d)
    return merged


# ---------------------------------------------------------------------------------------------- sharded scoring (configs[4])
def _all_gather_ranges(local, total, ranges, group=None):
    """Every rank contributes the 1-D tensor `local` = its [begin, end) slice of a vector of `total` elements
    (`ranges[r]` = rank r's slice); returns the whole vector on every rank.  One all-gather of equal-sized (padded) pieces."""
    if not _group_active(group):
        return local
    world = dist.get_world_size(group)
    width = max(e - b for b, e in ranges)
    if local.numel() == width:                                              # (no padding needed: the slice itself is the piece)
        piece = local.contiguous()
    else:
        piece = torch.zeros(width, dtype=local.dtype, device=local.device)
        piece[: local.numel()] = local
    if dist.get_backend(group) == "nccl":                                   # RCCL: device buffers, one collective
        out = torch.empty(world * width, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, piece, group=group)
    else:                                                                   # gloo (CPU tests; device tensors staged through the host)
        host = piece.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        out = torch.cat(parts).to(local.device)
    if all(b == r * width and (e - b == width or r == world - 1) for r, (b, e) in enumerate(ranges)):
        full = out[:total]                                                  # equal pieces back to back, only the last one short: already the vector
    else:
        full = torch.cat([out[r * width: r * width + (e - b)] for r, (b, e) in enumerate(ranges)])
    assert full.numel() == total
    return full


def sharded_hyperbolic_scores(n_windows, window, evaluate, kde_modes, finish, need_norms=False, group=None):
    """Window scores of the hyperbolic branch (utils/anomaly_detection_utils.py:54-86) with the windows split over the ranks
    (BASELINE.json configs[4]).  Every rank gets the full result.

    * rank r evaluates windows [b - (window - 1), e): its own range plus the halo whose critic values reach its un-rolled
      timesteps -- re-computed locally, not exchanged.  ``evaluate(lo, hi)`` -> dict of 1-D tensors over those windows:
      ``rowdist`` (row-wise Poincare distance), ``critic``, and ``norms`` (||recons||_2, only if ``need_norms``);
    * ``kde_modes(critic, window)`` -> the KDE mode of every un-rolled timestep of the evaluated windows
      (:374-400); the rank keeps the timesteps it owns;
    * one all-gather each of the (N,) distances [and norms] and the (N + window - 1,) modes -- 4-8 MB at 10^6 windows;
    * ``finish(rowdist, modes, norms)`` -> final scores: the global steps (quantile-trimmed z-score of the modes, rolling
      mean, combination) on the full vectors, exactly the unsharded code, so the result does not depend on the world size.
    """
    world = dist.get_world_size(group) if _group_active(group) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    b, e = window_range(n_windows, world, rank)
    hb, he = window_range_with_halo(n_windows, world, rank, window)
    tb, te = timestep_range(n_windows, world, rank, window)
    if e > b or rank == world - 1:
        got = evaluate(hb, he)
        modes = kde_modes(got["critic"], window)[tb - hb: te - hb]          # local timestep k == global hb + k
        rowdist = got["rowdist"][b - hb:]
        norms = got["norms"][b - hb:] if need_norms else None
    else:                                                                   # more ranks than windows: nothing owned
        ref = evaluate(0, min(1, n_windows))
        modes, rowdist = ref["critic"][:0].to(torch.float64), ref["rowdist"][:0]
        norms = ref["norms"][:0] if need_norms else None
    wr = [window_range(n_windows, world, r) for r in range(world)]
    tr = [timestep_range(n_windows, world, r, window) for r in range(world)]
    rowdist = _all_gather_ranges(rowdist.contiguous(), n_windows, wr, group)
    modes = _all_gather_ranges(modes.contiguous(), n_windows + window - 1, tr, group)
    if need_norms:
        norms = _all_gather_ranges(norms.contiguous(), n_windows, wr, group)
    return finish(rowdist, modes, norms)


_WS_CACHE = {}
_GRAPHS = {}


def replay_scorer(fn, *tensors, key=()):
    """A scorer call as ONE hipGraph: ``fn()`` -- a call of ``score_windows_sharded`` / ``score_anomalies_sharded`` with
    ``as_tensor=True`` on device-resident inputs -- is a fixed launch sequence (two large kernels and ~25 small ones, the
    collectives included; nothing passes through the host), so repeated scoring of same-shaped inputs can replay it instead of
    enqueueing it: captured on first use (after one eager call: scratch buffers, RCCL set-up), keyed by ``key`` and the addresses /
    shapes of ``tensors`` (the inputs and weights the capture froze; refill them in place between calls).  Returns the scores as
    the graph's own output tensor: it is overwritten by the next replay -- copy it to keep it."""
    k = (key,) + tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in tensors)
    ent = _GRAPHS.get(k)
    if ent is None:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        ent = _GRAPHS[k] = (g, out)
    ent[0].replay()
    return ent[1]


def _score_workspace(device, S, L, hyperbolic):
    """The fused forward's workspace (packed weights, rebuilt by every call of it), allocated once per (device, shape)."""
    from . import _C
    key = (str(device), S, L, int(hyperbolic))
    if key not in _WS_CACHE:
        nbytes = _C.lib.hypad_score_workspace_bytes(S, L, int(hyperbolic))
        _WS_CACHE[key] = (torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=device), nbytes)
    return _WS_CACHE[key]


def score_windows_sharded(x, encoder, decoder, critic_x, signal_shape, combination="mult", group=None, x_row_stride=0,
                          n_windows=None, as_tensor=False):
    """``sharded_hyperbolic_scores`` on the device kernels: ``x`` is the (N, S) fp32 window matrix -- or, with
    ``x_row_stride=1``, the scaled series whose window n is x[n : n + S] -- resident on every rank, like the weights
    (broadcast them once with ``broadcast_weights`` if the ranks did not load the same checkpoint).  Returns the final
    scores, (N,) float64, on every rank: a NumPy array as the reference's functions return, or (``as_tensor``) the device
    tensor.  Per window only its distance and critic value leave the fused forward (8 bytes; the reconstructions are
    written only where a combination needs their norms)."""
    import math
    from . import _C
    from .utils import anomaly_detection_utils as adu
    if not decoder.hyperbolic:
        raise ValueError("score_windows_sharded is the hyperbolic branch; use score_anomalies_sharded for Euclidean models")
    S, L = signal_shape, encoder.latent_space_dim
    N = n_windows if n_windows is not None else (x.shape[0] if x_row_stride == 0 else x.numel() - S + 1)
    need_norms = "uncertainty" in combination
    ws, ws_bytes = _score_workspace(x.device, S, L, 1)
    encoder.eval(); decoder.eval(); critic_x.eval()

    def evaluate(lo, hi):
        n = hi - lo
        new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        hyper = new(n, S) if need_norms else None
        critic, dist_ = new(n), new(n)
        xs = x[lo:hi] if x_row_stride == 0 else x[lo * x_row_stride:]
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(encoder.arena()), _C.ptr(decoder.arena()), _C.ptr(critic_x.arena()), _C.ptr(xs),
                                                   x_row_stride, _C.ptr(hyper), None, None, _C.ptr(critic), _C.ptr(dist_),
                                                   n, S, L, 1, ws.data_ptr(), ws_bytes, _C.stream()), "score_forward_packed")
        out = {"rowdist": dist_, "critic": critic}
        if need_norms:
            out["norms"] = adu.row_norms(hyper)
        return out

    def finish(rowdist, modes, norms):
        critic_scores = []
        if combination in ("mult", "uncertainty", "sum", "sum_uncertainty", "critic", "critic_uncertainty"):
            critic_scores = adu._compute_critic_score(modes, math.trunc(N * 0.01))[:N]
        return adu.combine_scores(combination, critic_scores, rowdist, norms=norms, as_tensor=as_tensor)

    return sharded_hyperbolic_scores(N, S, evaluate, adu.kde_modes, finish, need_norms, group)


# ---------------------------------------------------------------------------------------------- sharded Euclidean scoring
def error_halo(score_window=10):
    """Timesteps an error value reaches on either side.  `_area_error`'s centred window of `score_window` covers [t-5, t+4].
    `_dtw_error` (:834-861) pads by 5, compares y_pad[i : i+11] and files the result at position i + 5 -- so the value at t
    looks at true[t-10 .. t] (not a centred window), positions t < 5 and t >= T - 6 are framed with zeros, and t < 10 sees
    pad zeros.  length + 1 = 12 covers all of it."""
    return 2 * (score_window // 2) + 2



def extended_timestep_range(n_windows, world, rank, window, smooth_window, score_window=10):
    """Timesteps a rank must *compute* so that the smoothed error of every timestep it owns is exact: its own range widened
    by the reach of the error function and of the centred rolling mean (:953-961), clipped to the series."""
    tb, te = timestep_range(n_windows, world, rank, window)
    h = error_halo(score_window) + smooth_window // 2 + 1
    return max(0, tb - h), min(n_windows + window - 1, te + h)


def sharded_euclidean_scores(n_windows, window, smooth_window, evaluate, unroll_median, error_fn, rolling_mean, kde_modes, finish,
                             score_window=10, group=None, zscore="gather"):
    """Timestep scores of the Euclidean branch -- ``score_anomalies`` (utils/anomaly_detection_utils.py:407-576) -- with the
    windows split over the ranks (the DTW leg of BASELINE.json configs[4]).  Every rank gets the full result.

    rank r owns the un-rolled timesteps ``timestep_range`` gives it and computes, locally and without any exchange:
      * ``evaluate(lo, hi)`` on the windows that reach its *extended* timestep range (own range + error halo + rolling-mean
        halo, then the S-1 windows before it) -> dict(recon (n, S), critic (n,), true (n + S - 1,) = the un-rolled true
        series of those windows, :908-910);
      * ``unroll_median(recon)`` (:918-923), ``error_fn(true, pred)`` (point / area / DTW, :761-863) over the extended range
        as a series of its own -- only values whose whole footprint lies inside the range (or at a true end of the series,
        where the reference's own edge handling applies) are kept;
      * ``rolling_mean(err, smooth_window, origin)`` (:953-961) the same way -- ``origin`` = position of ``err[0]`` in the whole
        series: an implementation whose summation order depends on absolute positions (hypad_rolling_mean's chunked sums) then
        gives a slice the bits of the whole;  ``kde_modes(critic, window)`` for its timesteps.
    Exchanged: one all-gather each of the (T,) smoothed errors and (T,) critic modes -- 16 MB at 10^6 windows -- then
    ``finish(err, modes)`` runs the global steps (z-score / clip, quantile-trimmed critic z-score, combination) on the full
    vectors on every rank: the scores do not depend on the world size.  ``zscore="allreduce"`` instead normalises each
    rank's errors with the global (sum, sum of squares, count) of ONE 24-byte all-reduce before the gather (``finish`` then
    receives z-scores; the summation order, hence the last bits, depend on the world size)."""
    world = dist.get_world_size(group) if _group_active(group) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    T = n_windows + window - 1
    tb, te = timestep_range(n_windows, world, rank, window)
    a, b = extended_timestep_range(n_windows, world, rank, window, smooth_window, score_window)
    if te > tb:
        wlo, whi = max(0, a - (window - 1)), min(n_windows, b)
        got = evaluate(wlo, whi)
        pred = unroll_median(got["recon"])[a - wlo: b - wlo]            # local timestep k == global wlo + k
        true = got["true"][a - wlo: b - wlo]
        err = error_fn(true, pred)
        # keep what does not feel the artificial ends of the extended range
        h_err = error_halo(score_window)
        va, vb = (a + h_err if a > 0 else 0), (b - h_err if b < T else T)
        sm = rolling_mean(err[va - a: vb - a].contiguous(), smooth_window, va)      # va: where the slice sits in the whole series
        mine = sm[tb - va: te - va]
        # critic modes of the owned timesteps (they see windows [tb - S + 1, te), a subset of the evaluated ones)
        modes = kde_modes(got["critic"], window)[tb - wlo: te - wlo]
    else:
        ref = evaluate(0, min(1, n_windows))
        mine = ref["true"][:0].to(torch.float64)
        modes = ref["true"][:0].to(torch.float64)
    tr = [timestep_range(n_windows, world, r, window) for r in range(world)]
    if zscore == "allreduce":
        m64 = mine.to(torch.float64)
        mean, std = global_zscore_stats(float(m64.sum()), float((m64 * m64).sum()), int(m64.numel()), group)
        mine = (m64 - mean) / std
    elif zscore != "gather":
        raise ValueError(zscore)
    err_full = _all_gather_ranges(mine.contiguous(), T, tr, group)
    modes_full = _all_gather_ranges(modes.contiguous(), T, tr, group)
    return finish(err_full, modes_full)


def score_anomalies_sharded(y, encoder, decoder, critic_x, signal_shape, rec_error_type="point", comb="mult", score_window=10,
                            group=None, zscore="gather", as_tensor=False):
    """``sharded_euclidean_scores`` on the device kernels: the Euclidean branch of the reference's scoring
    (``test_tadgan`` batch body anomaly_detection.py:67-113 -> ``score_anomalies`` utils/anomaly_detection_utils.py:407-576)
    for a window matrix ``y`` (N, S) -- float32 or float64, resident on every rank like the weights.  Returns the final
    scores, (N + S - 1,) float64 -- NumPy, or (``as_tensor``) the device tensor -- on every rank; equal bit for bit to
    ``utils.anomaly_detection_utils.score_anomalies`` on the un-sharded reconstructions (``zscore="gather"``)."""
    import math
    from . import _C
    from .utils import anomaly_detection_utils as adu
    if decoder.hyperbolic:
        raise ValueError("score_anomalies_sharded is the Euclidean branch; use score_windows_sharded for hyperbolic models")
    S, L = signal_shape, encoder.latent_space_dim
    N = y.shape[0]
    y = y.reshape(N, S)
    w = math.trunc(N * 0.01)
    kind = rec_error_type.lower()
    if kind not in ("point", "area", "dtw"):
        raise ValueError(rec_error_type)
    ws, ws_bytes = _score_workspace(y.device, S, L, 0)
    encoder.eval(); decoder.eval(); critic_x.eval()

    def evaluate(lo, hi):
        n = hi - lo
        xs = y[lo:hi].to(torch.float32).contiguous()            # (no copy when the windows already are fp32)
        recon = torch.empty(n, S, device=y.device, dtype=torch.float32)
        critic = torch.empty(n, device=y.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(encoder.arena()), _C.ptr(decoder.arena()), _C.ptr(critic_x.arena()), _C.ptr(xs), 0,
                                                   None, _C.ptr(recon), None, _C.ptr(critic), None, n, S, L, 0, ws.data_ptr(), ws_bytes,
                                                   _C.stream()), "score_forward_packed")
        return {"recon": recon, "critic": critic, "true": adu.unroll_true(y[lo:hi])}

    def error_fn(true, pred):
        true, pred = true.contiguous(), pred.contiguous()
        if kind == "point":
            return adu._point_wise_error(true, pred)
        return (adu._area_error if kind == "area" else adu._dtw_error)(true, pred, score_window)

    def finish(err, modes):
        rec_scores = err if zscore == "allreduce" else None
        if rec_scores is None:
            rec_scores = adu.zscore_clip(err)
        else:
            rec_scores = torch.clamp(rec_scores, min=0) + 1
        critic_scores = adu._compute_critic_score(modes, w)
        return adu.combine_euclidean(comb, critic_scores, rec_scores, as_tensor=as_tensor)

    return sharded_euclidean_scores(N, S, w, evaluate, lambda r: adu.unroll_predictions(r, False)[0], error_fn, adu.rolling_mean,
                                    adu.kde_modes, finish, score_window, group, zscore)
